// user_env.h -- Env<TFMPC_ENV_USER>: "any differentiable env" (the reference differentiates whatever transition / cost it is handed,
// tfmpc/envs/diffenv.py:13-101) as DEVICE code at hot-path speed.  The user writes three device functions, templated on the scalar type:
//
//     template <class S> __device__ void transition(const float *p, const S *x, const S *u, S *x_next);   // x_next[TFMPC_USER_N]
//     template <class S> __device__ S    cost(const float *p, const S *x, const S *u);
//     template <class S> __device__ S    final_cost(const float *p, const S *x);
//
// (`p`: the instance's parameter floats; arithmetic, comparisons, sqrt exp log sin cos tanh abs pow max min on S.)  tfmpc.envs.deviceenv
// wraps them into a translation unit (user_env_kernels.hip.in) that is compiled with hipcc when the env is first used and instantiates the
// SAME wave-per-instance kernels the built-in envs run (ilqr_wave_kernels.h) on this Env.  The derivatives iLQR needs
// (ilqr.py:84-92: f_x, f_u, l_x, l_u, l_xx, l_uu, l_ux; final l_x, l_xx) come from FORWARD-mode automatic differentiation, one direction per
// LANE: lane j evaluates `transition` on dual numbers seeded with e_j and holds column j of [f_x | f_u]; lane (i, j) evaluates `cost` on
// second-order duals seeded with (e_i, e_j) and holds l_i, l_j and l_ij -- so a whole linearisation costs about one evaluation of each
// function, all 64 lanes busy.  Semantics of the reference's GradientTape.batch_jacobian with unconnected_gradients=ZERO
// (diffenv.py:21-24,40-72): an output that does not depend on an input has derivative 0; max / min / abs differentiate like TensorFlow's
// (ties of max(a, b) go to the FIRST argument, |y|' = sign(y) with 0 at 0: SURVEY.md Appendix A.3).
//
// Included TWICE by the translation unit (user_env_kernels.hip.in): first for the dual numbers (before the user's source, which may name
// them), then -- with TFMPC_USER_ENV_DEFINE, TFMPC_USER_N, TFMPC_USER_M and the three functions in namespace tfmpc_user defined -- for the Env.
#ifndef TFMPC_USER_ENV_AD_H
#define TFMPC_USER_ENV_AD_H

#include <hip/hip_runtime.h>

#include "envs.h"

namespace tfmpc {
namespace ad {

// value + one infinitesimal direction; nested once (Dual<Dual<float>>) it carries two directions and their mixed second derivative
template <class T> struct Dual {
    T v, d;
    __device__ Dual() {}
    __device__ Dual(float c) : v(c), d(0.0f) {}
    __device__ Dual(T v_, T d_) : v(v_), d(d_) {}
};

__device__ __forceinline__ float prim(float x) { return x; }
template <class T> __device__ __forceinline__ float prim(const Dual<T> &a) { return prim(a.v); }

template <class T> __device__ __forceinline__ Dual<T> operator+(const Dual<T> &a, const Dual<T> &b) { return Dual<T>(a.v + b.v, a.d + b.d); }
template <class T> __device__ __forceinline__ Dual<T> operator-(const Dual<T> &a, const Dual<T> &b) { return Dual<T>(a.v - b.v, a.d - b.d); }
template <class T> __device__ __forceinline__ Dual<T> operator*(const Dual<T> &a, const Dual<T> &b) { return Dual<T>(a.v * b.v, a.d * b.v + a.v * b.d); }
template <class T> __device__ __forceinline__ Dual<T> operator/(const Dual<T> &a, const Dual<T> &b)
{
    const T q = a.v / b.v;
    return Dual<T>(q, (a.d - q * b.d) / b.v);
}
template <class T> __device__ __forceinline__ Dual<T> operator-(const Dual<T> &a) { return Dual<T>(-a.v, -a.d); }
template <class T> __device__ __forceinline__ Dual<T> operator+(const Dual<T> &a) { return a; }
// mixed with plain floats (parameters, literals)
template <class T> __device__ __forceinline__ Dual<T> operator+(const Dual<T> &a, float b) { return Dual<T>(a.v + b, a.d); }
template <class T> __device__ __forceinline__ Dual<T> operator+(float a, const Dual<T> &b) { return Dual<T>(a + b.v, b.d); }
template <class T> __device__ __forceinline__ Dual<T> operator-(const Dual<T> &a, float b) { return Dual<T>(a.v - b, a.d); }
template <class T> __device__ __forceinline__ Dual<T> operator-(float a, const Dual<T> &b) { return Dual<T>(a - b.v, -b.d); }
template <class T> __device__ __forceinline__ Dual<T> operator*(const Dual<T> &a, float b) { return Dual<T>(a.v * b, a.d * b); }
template <class T> __device__ __forceinline__ Dual<T> operator*(float a, const Dual<T> &b) { return Dual<T>(a * b.v, a * b.d); }
template <class T> __device__ __forceinline__ Dual<T> operator/(const Dual<T> &a, float b) { return Dual<T>(a.v / b, a.d / b); }
template <class T> __device__ __forceinline__ Dual<T> operator/(float a, const Dual<T> &b)
{
    const T q = a / b.v;
    return Dual<T>(q, -(q * b.d) / b.v);
}
template <class T> __device__ __forceinline__ Dual<T> &operator+=(Dual<T> &a, const Dual<T> &b) { a = a + b; return a; }
template <class T> __device__ __forceinline__ Dual<T> &operator-=(Dual<T> &a, const Dual<T> &b) { a = a - b; return a; }
template <class T> __device__ __forceinline__ Dual<T> &operator*=(Dual<T> &a, const Dual<T> &b) { a = a * b; return a; }
template <class T> __device__ __forceinline__ Dual<T> &operator/=(Dual<T> &a, const Dual<T> &b) { a = a / b; return a; }
template <class T> __device__ __forceinline__ Dual<T> &operator+=(Dual<T> &a, float b) { a = a + b; return a; }
template <class T> __device__ __forceinline__ Dual<T> &operator-=(Dual<T> &a, float b) { a = a - b; return a; }
template <class T> __device__ __forceinline__ Dual<T> &operator*=(Dual<T> &a, float b) { a = a * b; return a; }
template <class T> __device__ __forceinline__ Dual<T> &operator/=(Dual<T> &a, float b) { a = a / b; return a; }
// comparisons look at the value (the branch a float program would take)
#define TFMPC_AD_CMP(OP)                                                                                                          \
    template <class T> __device__ __forceinline__ bool operator OP(const Dual<T> &a, const Dual<T> &b) { return prim(a) OP prim(b); } \
    template <class T> __device__ __forceinline__ bool operator OP(const Dual<T> &a, float b) { return prim(a) OP b; }            \
    template <class T> __device__ __forceinline__ bool operator OP(float a, const Dual<T> &b) { return a OP prim(b); }
TFMPC_AD_CMP(<) TFMPC_AD_CMP(>) TFMPC_AD_CMP(<=) TFMPC_AD_CMP(>=) TFMPC_AD_CMP(==) TFMPC_AD_CMP(!=)
#undef TFMPC_AD_CMP

// elementary functions: f(a) = f(v) + f'(v) d, written on T so that they nest.  (User code calls sqrt / exp / ... unqualified: argument-
// dependent lookup finds the overloads below for duals and HIP's own float overloads for floats; in here the float case goes through the
// underscore names, because inside this namespace the templates would hide the global functions.)
__device__ __forceinline__ float sqrt_(float x) { return ::sqrtf(x); }
__device__ __forceinline__ float exp_(float x) { return ::expf(x); }
__device__ __forceinline__ float log_(float x) { return ::logf(x); }
__device__ __forceinline__ float sin_(float x) { return ::sinf(x); }
__device__ __forceinline__ float cos_(float x) { return ::cosf(x); }
__device__ __forceinline__ float tanh_(float x) { return ::tanhf(x); }
__device__ __forceinline__ float tan_(float x) { return ::tanf(x); }
__device__ __forceinline__ float atan_(float x) { return ::atanf(x); }
__device__ __forceinline__ float asin_(float x) { return ::asinf(x); }
__device__ __forceinline__ float acos_(float x) { return ::acosf(x); }
__device__ __forceinline__ float sinh_(float x) { return ::sinhf(x); }
__device__ __forceinline__ float cosh_(float x) { return ::coshf(x); }
__device__ __forceinline__ float erf_(float x) { return ::erff(x); }
__device__ __forceinline__ float atan2_(float y, float x) { return ::atan2f(y, x); }
__device__ __forceinline__ float pow_(float x, float p) { return ::powf(x, p); }
__device__ __forceinline__ float abs_(float x) { return ::fabsf(x); }
template <class T> __device__ __forceinline__ Dual<T> sqrt(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> exp(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> log(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> sin(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> cos(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> tanh(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> pow(const Dual<T> &a, float p);
template <class T> __device__ __forceinline__ Dual<T> abs(const Dual<T> &a);
// (round 6: what a vehicle or arm model is made of -- tan, the inverse trigonometric functions incl. atan2, the hyperbolic pair, erf)
template <class T> __device__ __forceinline__ Dual<T> tan(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> atan(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> asin(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> acos(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> sinh(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> cosh(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> erf(const Dual<T> &a);
template <class T> __device__ __forceinline__ Dual<T> atan2(const Dual<T> &y, const Dual<T> &x);
template <class T> __device__ __forceinline__ Dual<T> tan_(const Dual<T> &a) { return tan(a); }
template <class T> __device__ __forceinline__ Dual<T> atan_(const Dual<T> &a) { return atan(a); }
template <class T> __device__ __forceinline__ Dual<T> asin_(const Dual<T> &a) { return asin(a); }
template <class T> __device__ __forceinline__ Dual<T> acos_(const Dual<T> &a) { return acos(a); }
template <class T> __device__ __forceinline__ Dual<T> sinh_(const Dual<T> &a) { return sinh(a); }
template <class T> __device__ __forceinline__ Dual<T> cosh_(const Dual<T> &a) { return cosh(a); }
template <class T> __device__ __forceinline__ Dual<T> erf_(const Dual<T> &a) { return erf(a); }
template <class T> __device__ __forceinline__ Dual<T> atan2_(const Dual<T> &y, const Dual<T> &x) { return atan2(y, x); }
template <class T> __device__ __forceinline__ Dual<T> sqrt_(const Dual<T> &a) { return sqrt(a); }
template <class T> __device__ __forceinline__ Dual<T> exp_(const Dual<T> &a) { return exp(a); }
template <class T> __device__ __forceinline__ Dual<T> log_(const Dual<T> &a) { return log(a); }
template <class T> __device__ __forceinline__ Dual<T> sin_(const Dual<T> &a) { return sin(a); }
template <class T> __device__ __forceinline__ Dual<T> cos_(const Dual<T> &a) { return cos(a); }
template <class T> __device__ __forceinline__ Dual<T> tanh_(const Dual<T> &a) { return tanh(a); }
template <class T> __device__ __forceinline__ Dual<T> pow_(const Dual<T> &a, float p) { return pow(a, p); }
template <class T> __device__ __forceinline__ Dual<T> abs_(const Dual<T> &a) { return abs(a); }
template <class T> __device__ __forceinline__ Dual<T> sqrt(const Dual<T> &a) { const T s = sqrt_(a.v); return Dual<T>(s, a.d / (2.0f * s)); }
template <class T> __device__ __forceinline__ Dual<T> exp(const Dual<T> &a) { const T e = exp_(a.v); return Dual<T>(e, a.d * e); }
template <class T> __device__ __forceinline__ Dual<T> log(const Dual<T> &a) { return Dual<T>(log_(a.v), a.d / a.v); }
template <class T> __device__ __forceinline__ Dual<T> sin(const Dual<T> &a) { return Dual<T>(sin_(a.v), a.d * cos_(a.v)); }
template <class T> __device__ __forceinline__ Dual<T> cos(const Dual<T> &a) { return Dual<T>(cos_(a.v), -(a.d * sin_(a.v))); }
template <class T> __device__ __forceinline__ Dual<T> tanh(const Dual<T> &a) { const T t = tanh_(a.v); return Dual<T>(t, a.d * (1.0f - t * t)); }
template <class T> __device__ __forceinline__ Dual<T> pow(const Dual<T> &a, float p) { return Dual<T>(pow_(a.v, p), a.d * (p * pow_(a.v, p - 1.0f))); }
template <class T> __device__ __forceinline__ Dual<T> tan(const Dual<T> &a) { const T t = tan_(a.v); return Dual<T>(t, a.d * (1.0f + t * t)); }
template <class T> __device__ __forceinline__ Dual<T> atan(const Dual<T> &a) { return Dual<T>(atan_(a.v), a.d / (1.0f + a.v * a.v)); }
template <class T> __device__ __forceinline__ Dual<T> asin(const Dual<T> &a) { return Dual<T>(asin_(a.v), a.d / sqrt_(1.0f - a.v * a.v)); }
template <class T> __device__ __forceinline__ Dual<T> acos(const Dual<T> &a) { return Dual<T>(acos_(a.v), -(a.d / sqrt_(1.0f - a.v * a.v))); }
template <class T> __device__ __forceinline__ Dual<T> sinh(const Dual<T> &a) { return Dual<T>(sinh_(a.v), a.d * cosh_(a.v)); }
template <class T> __device__ __forceinline__ Dual<T> cosh(const Dual<T> &a) { return Dual<T>(cosh_(a.v), a.d * sinh_(a.v)); }
template <class T> __device__ __forceinline__ Dual<T> erf(const Dual<T> &a) { return Dual<T>(erf_(a.v), a.d * (1.1283791670955126f * exp_(-(a.v * a.v)))); }
template <class T> __device__ __forceinline__ Dual<T> atan2(const Dual<T> &y, const Dual<T> &x)
{
    return Dual<T>(atan2_(y.v, x.v), (y.d * x.v - x.d * y.v) / (x.v * x.v + y.v * y.v));
}
template <class T> __device__ __forceinline__ Dual<T> atan2(const Dual<T> &y, float x) { return atan2(y, Dual<T>(x)); }
template <class T> __device__ __forceinline__ Dual<T> atan2(float y, const Dual<T> &x) { return atan2(Dual<T>(y), x); }
// |y|' = sign(y), 0 at 0; max / min: a tie goes to the first argument (TensorFlow's gradients: SURVEY.md Appendix A.3)
template <class T> __device__ __forceinline__ Dual<T> abs(const Dual<T> &a)
{
    const float y = prim(a);
    return y > 0.0f ? a : (y < 0.0f ? -a : Dual<T>(abs_(a.v), a.d * 0.0f));
}
template <class T> __device__ __forceinline__ Dual<T> fabs(const Dual<T> &a) { return abs(a); }
template <class T> __device__ __forceinline__ Dual<T> max(const Dual<T> &a, const Dual<T> &b) { return prim(a) >= prim(b) ? a : b; }
template <class T> __device__ __forceinline__ Dual<T> min(const Dual<T> &a, const Dual<T> &b) { return prim(a) <= prim(b) ? a : b; }
template <class T> __device__ __forceinline__ Dual<T> max(const Dual<T> &a, float b) { return prim(a) >= b ? a : Dual<T>(b); }
template <class T> __device__ __forceinline__ Dual<T> max(float a, const Dual<T> &b) { return a >= prim(b) ? Dual<T>(a) : b; }
template <class T> __device__ __forceinline__ Dual<T> min(const Dual<T> &a, float b) { return prim(a) <= b ? a : Dual<T>(b); }
template <class T> __device__ __forceinline__ Dual<T> min(float a, const Dual<T> &b) { return a <= prim(b) ? Dual<T>(a) : b; }
template <class T> __device__ __forceinline__ Dual<T> fmax(const Dual<T> &a, const Dual<T> &b) { return max(a, b); }
template <class T> __device__ __forceinline__ Dual<T> fmin(const Dual<T> &a, const Dual<T> &b) { return min(a, b); }

using D1 = Dual<float>;
using D2 = Dual<Dual<float>>;

}  // namespace ad
}  // namespace tfmpc

#endif      // TFMPC_USER_ENV_AD_H

#if defined(TFMPC_USER_ENV_DEFINE) && !defined(TFMPC_USER_ENV_DEFINED)
#define TFMPC_USER_ENV_DEFINED

namespace tfmpc {

// The instance's parameter floats are p[0] (TfmpcEnv::n_zones of them: envs.h:env_param_len), copied to LDS like every env's.
#ifndef TFMPC_USER_ZERO_HESSIAN
#define TFMPC_USER_ZERO_HESSIAN 0
#endif
template <> struct Env<TFMPC_ENV_USER> {
    // TFMPC_USER_ZERO_HESSIAN (stated by the Python side: tfmpc.envs.deviceenv.DeviceEnv(zero_cost_hessian=True), or proved by the translator of
    // TorchEnv.to_device_env): every second derivative of cost / final_cost is identically zero and the actions are bounded -- V_xx stays exactly 0,
    // the backward pass of ilqr.py:94-172 only ever takes its bang-bang branch (:137-141), and the fused solve kernel runs the COSTATE
    // form: `adjoint_direction` / `final_grad` below (one first-order dual evaluation per direction) and `speculative_search`.
    static constexpr bool kPiecewiseLinearCost = TFMPC_USER_ZERO_HESSIAN != 0;
    static constexpr int N = TFMPC_USER_N, M = TFMPC_USER_M, D = N + M;

    // direction j of z = [x; u]: Q_z[j] = dl/dz_j + sum_i df_i/dz_j V_x[i] (ilqr.py:122-123), ONE first-order dual evaluation of transition and cost
    // in this lane; returns the stage cost (every lane: the same value)
    static __device__ float adjoint_direction(const EnvLds &e, const float *x, const float *u, const float *Vx, int j, float &qz)
    {
        using ad::D1;
        D1 xs[N], us[M], out[N];
#pragma unroll
        for (int i = 0; i < N; ++i) xs[i] = D1(x[i], i == j ? 1.0f : 0.0f);
#pragma unroll
        for (int a = 0; a < M; ++a) us[a] = D1(u[a], N + a == j ? 1.0f : 0.0f);
        tfmpc_user::transition<D1>(e.p[0], xs, us, out);
        const D1 c = tfmpc_user::cost<D1>(e.p[0], xs, us);
        float acc = c.d;
#pragma unroll
        for (int i = 0; i < N; ++i) acc = fmaf(out[i].d, Vx[i], acc);
        qz = acc;
        return c.v;
    }
    // V_x = l_x^f (ilqr.py:101), lane j < N its entry j; returns the final cost
    static __device__ float final_grad(const EnvLds &e, const float *x, float *Vx)
    {
        using ad::D1;
        float value = 0.0f;
        for (int base = 0; base < N; base += kWave) {
            const int j = base + lane_id();
            D1 xs[N];
#pragma unroll
            for (int i = 0; i < N; ++i) xs[i] = D1(x[i], i == j ? 1.0f : 0.0f);
            const D1 c = tfmpc_user::final_cost<D1>(e.p[0], xs);
            if (j < N) Vx[j] = c.d;
            value = c.v;
        }
        return value;
    }
    // The line search of ilqr.py:317-355 with K == 0, EVERY step size at once, one per lane: the user's transition / cost are scalar programs that each
    // lane of a wave evaluates anyway (wave-uniform in the sequential rollout), so lane s rolls out step size s for the price of one rollout.  Lane `guess`
    // (the step size the previous pass accepted) stores its candidate as it goes; the reference's rule -- the FIRST step size with z >= c1, else the last
    // one, rejected (:322-353) -- is a ballot.  `chosen` != `guess`: the caller rolls the chosen one out again (forward_pass, same arithmetic: same bits).
    // `extra` (round 6; may be null): room for three more candidates of this instance, [x | u | c] each -- the lanes of the step sizes guess - 1 ..
    // guess + 2 then all store theirs (slot 1 = xc, uc, cc; the accepted index moves between passes: with the guess alone stored four passes of five
    // rolled the chosen step size out again, tools/probes/r6_alpha_moves.py).  `from_slot` <- the slot the chosen candidate is in, -1: not stored.
    static constexpr int kSlots = 4, kSlotOfGuess = 1;
    static __host__ __device__ size_t candidate_floats(int T) { return (size_t)(T + 1) * N + (size_t)T * M + (size_t)(T + 1); }
    static __device__ void slot_pointers(int slot, int T, float *xc, float *uc, float *cc, float *extra, float *&xs, float *&us, float *&cs)
    {
        if (slot == kSlotOfGuess || !extra) { xs = xc; us = uc; cs = cc; return; }
        xs = extra + (size_t)(slot - (slot > kSlotOfGuess ? 1 : 0)) * candidate_floats(T);
        us = xs + (size_t)(T + 1) * N;
        cs = us + (size_t)T * M;
    }
    static __device__ void speculative_search(const EnvLds &e, const TfmpcIlqrConfig &cfg, int T, const float *xhat, const float *uhat, const float *kg,
                                              float J_hat, float dV1, int guess, float *xc_, float *uc_, float *cc_, float *extra, int &chosen, bool &accept,
                                              float &J_out, float &residual_out, int &from_slot)
    {
        const int lane = lane_id();
        const int mine = lane < cfg.n_alphas ? lane : cfg.n_alphas - 1;
        const float alpha = cfg.alphas[mine];
        const int my_slot = lane - guess + kSlotOfGuess;
        const bool keep = lane < cfg.n_alphas && (extra ? (my_slot >= 0 && my_slot < kSlots) : lane == guess);
        float *xc, *uc, *cc;
        slot_pointers(keep ? my_slot : kSlotOfGuess, T, xc_, uc_, cc_, extra, xc, uc, cc);
        const float *p = e.p[0];
        float x[N], xn[N], u[M];
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] = xhat[i]; if (keep) xc[i] = x[i]; }
        float J = 0.0f, rmax = 0.0f;
        // (round 6: the inputs of step t + 1 are requested while step t is evaluated -- a register ring of depth one, refilled unconditionally with a
        // clamped index; on the sixteen-lanes-per-instance form of this search, user_env_group.h, that was 17 - 29 % of a solve)
        float uh_n[M], k_n[M];
#pragma unroll
        for (int a = 0; a < M; ++a) { uh_n[a] = T > 0 ? uhat[a] : 0.0f; k_n[a] = T > 0 ? kg[a] : 0.0f; }
        for (int t = 0; t < T; ++t) {
            const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
            for (int a = 0; a < M; ++a) {
                const float uh_c = uh_n[a], k_c = k_n[a];
                uh_n[a] = uhat[(size_t)tn * M + a];
                k_n[a] = kg[(size_t)tn * M + a];
                const float du = alpha * k_c;                                                            // :193-194 (K == 0)
                u[a] = fminf(fmaxf(uh_c + du, e.low[a]), e.high[a]);                                     // :196-197
                rmax = fmaxf(rmax, fabsf(du));                                                            // :206
                if (keep) uc[(size_t)t * M + a] = u[a];
            }
            const float c = tfmpc_user::cost<float>(p, x, u);                                            // :198
            tfmpc_user::transition<float>(p, x, u, xn);                                                  // :199
            J += c;                                                                                      // :205
            if (keep) cc[t] = c;
#pragma unroll
            for (int i = 0; i < N; ++i) { x[i] = xn[i]; if (keep) xc[(size_t)(t + 1) * N + i] = xn[i]; }
        }
        const float fc = tfmpc_user::final_cost<float>(p, x);                                            // :208-210
        if (keep) cc[T] = fc;
        J += fc;
        const float delta_J = -alpha * (dV1 + alpha * 0.0f);                                             // :339 (dV2 == 0)
        const float dcost = J_hat - J;
        const float z = (delta_J > 0.0f) ? dcost / delta_J : ((dcost > 0.0f) ? 1.0f : ((dcost < 0.0f) ? -1.0f : 0.0f));   // :342-346
        const unsigned long long pass = __ballot(lane < cfg.n_alphas && z >= cfg.c1);
        accept = pass != 0ull;
        chosen = accept ? __builtin_ctzll(pass) : cfg.n_alphas - 1;
        J_out = __shfl(J, chosen, kWave);
        residual_out = __shfl(rmax, chosen, kWave);
        const int slot = chosen - guess + kSlotOfGuess;
        from_slot = extra ? ((slot >= 0 && slot < kSlots) ? slot : -1) : (chosen == guess ? kSlotOfGuess : -1);
    }

    template <class S>
    static __device__ __forceinline__ void seed_point(const float *x, const float *u, S (&xs)[N], S (&us)[M])
    {
#pragma unroll
        for (int i = 0; i < N; ++i) xs[i] = S(x[i]);
#pragma unroll
        for (int a = 0; a < M; ++a) us[a] = S(u ? u[a] : 0.0f);
    }

    // every lane evaluates the same float program (wave-uniform results); lane 0 stores
    static __device__ void transition(const EnvLds &e, const float *x, const float *u, float *xn)
    {
        float xs[N], us[M], out[N];
        seed_point(x, u, xs, us);
        tfmpc_user::transition<float>(e.p[0], xs, us, out);
        if (lane_id() == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) xn[i] = out[i];
        }
    }
    static __device__ float cost(const EnvLds &e, const float *x, const float *u)
    {
        float xs[N], us[M];
        seed_point(x, u, xs, us);
        return tfmpc_user::cost<float>(e.p[0], xs, us);
    }
    static __device__ float final_cost(const EnvLds &e, const float *x)
    {
        float xs[N], us[M];
        seed_point(x, nullptr, xs, us);
        return tfmpc_user::final_cost<float>(e.p[0], xs);
    }

    // pair index -> (i, j), i <= j < d, row-major over the upper triangle
    static __device__ __forceinline__ void unpair(int pidx, int d, int &i, int &j)
    {
        i = 0;
        int rem = pidx;
        while (rem >= d - i) { rem -= d - i; ++i; }
        j = i + rem;
    }

    static __device__ float linearize(const EnvLds &e, const float *x, const float *u, float *fx, float *fu,
                                      float *lx, float *lu, float *lxx, float *luu, float *lux)
    {
        using ad::D1;
        using ad::D2;
        const int lane = lane_id(), ldn = odd_ld(N), ldm = odd_ld(M);
        const float *p = e.p[0];
        // [f_x | f_u]: lane j holds column j (diffenv.py:21-24)
        for (int base = 0; base < D; base += kWave) {
            const int j = base + lane;
            D1 xs[N], us[M], out[N];
#pragma unroll
            for (int i = 0; i < N; ++i) xs[i] = D1(x[i], i == j ? 1.0f : 0.0f);
#pragma unroll
            for (int a = 0; a < M; ++a) us[a] = D1(u[a], N + a == j ? 1.0f : 0.0f);
            tfmpc_user::transition<D1>(p, xs, us, out);
            if (j < N) {
#pragma unroll
                for (int i = 0; i < N; ++i) fx[i * ldn + j] = out[i].d;
            } else if (j < D) {
#pragma unroll
                for (int i = 0; i < N; ++i) fu[i * ldm + (j - N)] = out[i].d;
            }
        }
        // gradient and Hessian of the stage cost: lane (i, j) holds l_i, l_j, l_ij (diffenv.py:40-72)
        float l = 0.0f;
        constexpr int P = D * (D + 1) / 2;
        for (int base = 0; base < P; base += kWave) {
            int i, j;
            unpair(base + lane < P ? base + lane : P - 1, D, i, j);
            D2 xs[N], us[M];
#pragma unroll
            for (int k = 0; k < N; ++k) xs[k] = D2(D1(x[k], k == i ? 1.0f : 0.0f), D1(k == j ? 1.0f : 0.0f, 0.0f));
#pragma unroll
            for (int a = 0; a < M; ++a) us[a] = D2(D1(u[a], N + a == i ? 1.0f : 0.0f), D1(N + a == j ? 1.0f : 0.0f, 0.0f));
            const D2 c = tfmpc_user::cost<D2>(p, xs, us);
            l = c.v.v;
            if (base + lane < P) {
                const float h = c.d.d;
                if (j < N) { lxx[i * ldn + j] = h; lxx[j * ldn + i] = h; }
                else if (i >= N) { luu[(i - N) * ldm + (j - N)] = h; luu[(j - N) * ldm + (i - N)] = h; }
                else lux[(j - N) * ldn + i] = h;                                  // l_ux[a][i] = d2 l / du_a dx_i
                if (i == j) { if (i < N) lx[i] = c.v.d; else lu[i - N] = c.v.d; }
            }
        }
        return l;
    }

    static __device__ float final_quad(const EnvLds &e, const float *x, float *lx, float *lxx)
    {
        using ad::D1;
        using ad::D2;
        const int lane = lane_id(), ldn = odd_ld(N);
        float l = 0.0f;
        constexpr int P = N * (N + 1) / 2;
        for (int base = 0; base < P; base += kWave) {
            int i, j;
            unpair(base + lane < P ? base + lane : P - 1, N, i, j);
            D2 xs[N];
#pragma unroll
            for (int k = 0; k < N; ++k) xs[k] = D2(D1(x[k], k == i ? 1.0f : 0.0f), D1(k == j ? 1.0f : 0.0f, 0.0f));
            const D2 c = tfmpc_user::final_cost<D2>(e.p[0], xs);
            l = c.v.v;
            if (base + lane < P) {
                lxx[i * ldn + j] = c.d.d;
                lxx[j * ldn + i] = c.d.d;
                if (i == j) lx[i] = c.v.d;
            }
        }
        return l;
    }
};

}  // namespace tfmpc

// ---- tiny user envs (n + m <= 4, user_env_kernels.hip.in: TFMPC_USER_LANE_GROUP): the lane-group kernel (16 lanes per instance, all step sizes of the line search at once, persistent groups +
// instance queue: ilqr_lane_kernels.h) on LaneEnv<TFMPC_ENV_USER>.  One lane evaluates the user's functions as ordinary scalar code; the whole
// quadratic model of a timestep is the "precomputed part" of the linearisation (kPre floats), which the kernel evaluates for all timesteps at once,
// one per lane, whenever the nominal trajectory changes -- D dual evaluations of `transition` and D (D + 1) / 2 second-order ones of `cost` each.
#ifdef TFMPC_USER_LANE_GROUP
#include "ilqr_lane_kernels.h"

namespace tfmpc {

template <int N, int M>
struct LaneEnv<TFMPC_ENV_USER, N, M> {
    static constexpr int D = N + M;
    static constexpr int kFx = 0, kFu = kFx + N * N, kLx = kFu + N * M, kLu = kLx + N, kLxx = kLu + M, kLuu = kLxx + N * N,
                         kLux = kLuu + M * M, kL = kLux + M * N, kPre = kL + 1;
    // A small parameter vector (TFMPC_USER_P <= 16 floats, stated by the Python side) is read ONCE per instance into registers: through the pointer
    // every use is a vector load with its wait (the env object lives in vector registers) -- in every step of every rollout (round 5; the built-in
    // Navigation env had the same habit, ilqr_lane_kernels.h).  Constant indices in the user's source then cost nothing.
#ifndef TFMPC_USER_P
#define TFMPC_USER_P 0
#endif
    static constexpr int kP = (TFMPC_USER_P > 0 && TFMPC_USER_P <= 16) ? TFMPC_USER_P : 0;
    const float *pg;
    float pr[kP > 0 ? kP : 1];
    __device__ void load(const TfmpcEnv &g, int b)
    {
        pg = g.p[0] + (size_t)b * g.stride[0];
#pragma unroll
        for (int i = 0; i < kP; ++i) pr[i] = pg[i];
    }
    __device__ const float *params() const { return kP > 0 ? pr : pg; }
    __device__ void transition(const float *x, const float *u, float *xn) const { tfmpc_user::transition<float>(params(), x, u, xn); }
    __device__ float cost(const float *x, const float *u) const { return tfmpc_user::cost<float>(params(), x, u); }
    __device__ float final_cost(const float *x) const { return tfmpc_user::final_cost<float>(params(), x); }
    __device__ void prelinearize(const float *x, const float *u, float *pre) const
    {
        using ad::D1;
        using ad::D2;
#pragma unroll
        for (int j = 0; j < D; ++j) {                                   // column j of [f_x | f_u]
            D1 xs[N], us[M], out[N];
#pragma unroll
            for (int i = 0; i < N; ++i) xs[i] = D1(x[i], i == j ? 1.0f : 0.0f);
#pragma unroll
            for (int a = 0; a < M; ++a) us[a] = D1(u[a], N + a == j ? 1.0f : 0.0f);
            tfmpc_user::transition<D1>(params(), xs, us, out);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (j < N) pre[kFx + i * N + j] = out[i].d;
                else pre[kFu + i * M + (j - N)] = out[i].d;
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
#pragma unroll
            for (int j = i; j < D; ++j) {                               // l_i, l_j, l_ij
                D2 xs[N], us[M];
#pragma unroll
                for (int k = 0; k < N; ++k) xs[k] = D2(D1(x[k], k == i ? 1.0f : 0.0f), D1(k == j ? 1.0f : 0.0f, 0.0f));
#pragma unroll
                for (int a = 0; a < M; ++a) us[a] = D2(D1(u[a], N + a == i ? 1.0f : 0.0f), D1(N + a == j ? 1.0f : 0.0f, 0.0f));
                const D2 c = tfmpc_user::cost<D2>(params(), xs, us);
                const float h = c.d.d;
                if (j < N) { pre[kLxx + i * N + j] = h; pre[kLxx + j * N + i] = h; }
                else if (i >= N) { pre[kLuu + (i - N) * M + (j - N)] = h; pre[kLuu + (j - N) * M + (i - N)] = h; }
                else pre[kLux + (j - N) * N + i] = h;
                if (i == j) { if (i < N) pre[kLx + i] = c.v.d; else pre[kLu + i - N] = c.v.d; }
                if (i == 0 && j == 0) pre[kL] = c.v.v;
            }
        }
    }
    __device__ void linearize_pre(const float *pre, const float *, const float *, LaneModel<N, M> &md) const
    {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            md.lx[i] = pre[kLx + i];
#pragma unroll
            for (int j = 0; j < N; ++j) { md.fx(i, j) = pre[kFx + i * N + j]; md.lxx(i, j) = pre[kLxx + i * N + j]; }
#pragma unroll
            for (int a = 0; a < M; ++a) md.fu(i, a) = pre[kFu + i * M + a];
        }
#pragma unroll
        for (int a = 0; a < M; ++a) {
            md.lu[a] = pre[kLu + a];
#pragma unroll
            for (int c = 0; c < M; ++c) md.luu(a, c) = pre[kLuu + a * M + c];
#pragma unroll
            for (int j = 0; j < N; ++j) md.lux(a, j) = pre[kLux + a * N + j];
        }
        md.l = pre[kL];
    }
    __device__ void linearize(const float *x, const float *u, LaneModel<N, M> &md) const
    {
        float pre[kPre];
        prelinearize(x, u, pre);
        linearize_pre(pre, x, u, md);
    }
    __device__ float final_quad(const float *x, float *lx, small::Mat<N, N> &lxx) const
    {
        using ad::D1;
        using ad::D2;
        float l = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = i; j < N; ++j) {
                D2 xs[N];
#pragma unroll
                for (int k = 0; k < N; ++k) xs[k] = D2(D1(x[k], k == i ? 1.0f : 0.0f), D1(k == j ? 1.0f : 0.0f, 0.0f));
                const D2 c = tfmpc_user::final_cost<D2>(params(), xs);
                lxx(i, j) = c.d.d;
                lxx(j, i) = c.d.d;
                if (i == j) lx[i] = c.v.d;
                l = c.v.v;
            }
        }
        return l;
    }
};

}  // namespace tfmpc
#endif      // 2 x 2

#endif      // TFMPC_USER_ENV_DEFINE
