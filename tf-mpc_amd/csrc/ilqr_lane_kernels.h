// ilqr_lane_kernels.h -- the lane-per-instance and 16-lanes-per-instance ("lane group") iLQR kernels for tiny envs as templates over the env
// kind: per-lane env models (LaneEnv<KIND, N, M>), the in-register box-QP, backward / forward passes and the two solve kernels.  Device code
// only: instantiated by the library (ilqr_lane.hip: Navigation, NavigationLQR at n = m = 2) and by the translation unit a DeviceEnv is compiled
// into (user_env_kernels.hip.in: LaneEnv<TFMPC_ENV_USER, 2, 2> of user_env.h).  See ilqr_lane.hip for the design notes.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "envs.h"
#include "ilqr_trace.h"
#include "small_linalg.h"

namespace tfmpc {

using small::Mat;

__device__ __forceinline__ float sgnf(float y) { return (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f); }

// ---- per-lane env models (same closed forms as envs.h) ---------------------------------
template <int N, int M>
struct LaneModel {                 // quadratic expansion at one (x, u)
    Mat<N, N> fx, lxx;
    Mat<N, M> fu;
    Mat<M, M> luu;
    Mat<M, N> lux;
    float lx[N], lu[M], l;
};

// 1 / x to ~1 ulp: the hardware reciprocal and one Newton step (three instructions; an IEEE division is ten, and a
// lane kernel is bound by the LENGTH of its dependent instruction stream, ~10 cycles per instruction)
__device__ __forceinline__ float lane_rcp(float x) { return env_rcp(x); }       // envs.h

template <int KIND, int N, int M> struct LaneEnv;

template <int N>
struct LaneEnv<TFMPC_ENV_NAVLQR, N, N> {                 // envs/lqr/navigation/__init__.py:30-47
    float goal[N], beta;
    __device__ void load(const TfmpcEnv &g, int b)
    {
#pragma unroll
        for (int i = 0; i < N; ++i) goal[i] = g.p[0][(size_t)b * g.stride[0] + i];
        beta = g.scalar[0];
    }
    __device__ void transition(const float *x, const float *u, float *xn) const
    {
#pragma unroll
        for (int i = 0; i < N; ++i) xn[i] = x[i] + u[i];
    }
    __device__ float final_cost(const float *x) const
    {
        float c1 = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) { const float dx = x[i] - goal[i]; c1 += dx * dx; }
        return c1;
    }
    __device__ float cost(const float *x, const float *u) const
    {
        float c2 = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) c2 += u[i] * u[i];
        return final_cost(x) + beta * c2;
    }
    // nothing of the linearisation is worth precomputing here (see LaneEnv<NAVIGATION>)
    static constexpr int kPre = 0;
    __device__ void prelinearize(const float *, const float *, float *) const {}
    __device__ void linearize_pre(const float *, const float *x, const float *u, LaneModel<N, N> &md) const { linearize(x, u, md); }
    __device__ void linearize(const float *x, const float *u, LaneModel<N, N> &md) const
    {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            md.lx[i] = 2.0f * (x[i] - goal[i]);
            md.lu[i] = 2.0f * beta * u[i];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float id = (i == j) ? 1.0f : 0.0f;
                md.fx(i, j) = id; md.fu(i, j) = id;
                md.lxx(i, j) = 2.0f * id; md.luu(i, j) = 2.0f * beta * id; md.lux(i, j) = 0.0f;
            }
        }
        md.l = cost(x, u);
    }
    __device__ float final_quad(const float *x, float *lx, Mat<N, N> &lxx) const
    {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            lx[i] = 2.0f * (x[i] - goal[i]);
#pragma unroll
            for (int j = 0; j < N; ++j) lxx(i, j) = (i == j) ? 2.0f : 0.0f;
        }
        return final_cost(x);
    }
};

template <int N>
struct LaneEnv<TFMPC_ENV_NAVIGATION, N, N> {             // envs/navigation/__init__.py:34-74
    float goal[N];
    const float *center, *decay;   // shared by the batch
    int zones;
    // Round 5: the first kZoneRegs zones' centres and decays are read ONCE per instance into registers.  Read where they are used they were two or
    // three vector loads -- a round trip to the cache each, with its wait -- per zone and evaluation: in every step of every rollout, and six
    // evaluations per step of the linearisation (the env object lives in vector registers, so the "uniform" addresses were not scalar loads).
#ifdef TFMPC_NAV_ZONES_FROM_MEMORY     // A/B builds
    static constexpr int kZoneRegs = 0;
#else
    static constexpr int kZoneRegs = 4;
#endif
    float zc[kZoneRegs > 0 ? kZoneRegs : 1][N], zd[kZoneRegs > 0 ? kZoneRegs : 1];
    __device__ void load(const TfmpcEnv &g, int b)
    {
#pragma unroll
        for (int i = 0; i < N; ++i) goal[i] = g.p[0][(size_t)b * g.stride[0] + i];
        center = g.p[1]; decay = g.p[2]; zones = g.n_zones;
#pragma unroll
        for (int z = 0; z < kZoneRegs; ++z) {
            zd[z] = z < zones ? decay[z] : 0.0f;
#pragma unroll
            for (int i = 0; i < N; ++i) zc[z][i] = z < zones ? center[z * N + i] : 0.0f;
        }
    }
    __device__ float zone_lambda(const float *x, int z, float *r_out, float *ex_out) const
    {
        float c[N], dz = 0.0f;
        bool found = false;
#pragma unroll
        for (int k = 0; k < kZoneRegs; ++k)                // (a select chain over the resident zones: `z` is wave-uniform, the compiler keeps it scalar)
            if (z == k) {
                found = true;
                dz = zd[k];
#pragma unroll
                for (int i = 0; i < N; ++i) c[i] = zc[k][i];
            }
        if (!found) {
            dz = decay[z];
#pragma unroll
            for (int i = 0; i < N; ++i) c[i] = center[z * N + i];
        }
        float r2 = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) { const float d = x[i] - c[i]; r2 = fmaf(d, d, r2); }
        return nav_zone_lambda(r2, dz, r_out, ex_out);                // envs.h: the same expression in every kernel
    }
    __device__ float zone_decay(int z) const
    {
        float dz = 0.0f;
        bool found = false;
#pragma unroll
        for (int k = 0; k < kZoneRegs; ++k)
            if (z == k) { found = true; dz = zd[k]; }
        return found ? dz : decay[z];
    }
    __device__ float zone_center(int z, int i) const
    {
        float v = 0.0f;
        bool found = false;
#pragma unroll
        for (int k = 0; k < kZoneRegs; ++k)
            if (z == k) { found = true; v = zc[k][i]; }
        return found ? v : center[z * N + i];
    }
    __device__ float deceleration(const float *x, float *grad) const
    {
        float lam = 1.0f;
        for (int z = 0; z < zones; ++z) lam *= zone_lambda(x, z, nullptr, nullptr);
        if (grad) {
#pragma unroll
            for (int i = 0; i < N; ++i) grad[i] = 0.0f;
            for (int z = 0; z < zones; ++z) {
                float r, ex;
                zone_lambda(x, z, &r, &ex);
                const float h = nav_zone_slope(zone_decay(z), ex);
                float others = 1.0f;
                for (int y = 0; y < zones; ++y)
                    if (y != z) others *= zone_lambda(x, y, nullptr, nullptr);
#pragma unroll
                for (int i = 0; i < N; ++i) grad[i] += h * (x[i] - zone_center(z, i)) * env_rcp(r) * others;
            }
        }
        return lam;
    }
    __device__ void transition(const float *x, const float *u, float *xn) const
    {
        const float lam = deceleration(x, nullptr);
#pragma unroll
        for (int i = 0; i < N; ++i) xn[i] = fmaf(lam, u[i], x[i]);
    }
    __device__ float final_cost(const float *x) const
    {
        float c1 = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) { const float dx = x[i] - goal[i]; c1 += dx * dx; }
        return c1;
    }
    __device__ float cost(const float *x, const float *) const { return final_cost(x); }
    // The deceleration factor and its gradient depend on x alone and are most of an iteration's backward sweep
    // (sqrt, exp and divisions per zone): pre = {lambda, grad lambda}.  linearize == prelinearize + linearize_pre,
    // the same expressions in the same order, so a kernel that evaluates the first part elsewhere (the group kernel:
    // all timesteps at once, one per lane) gets the same bits.
    static constexpr int kPre = N + 1;
    __device__ void prelinearize(const float *x, const float *, float *pre) const
    {
        float grad[N];
        pre[0] = deceleration(x, grad);
#pragma unroll
        for (int i = 0; i < N; ++i) pre[1 + i] = grad[i];
    }
    __device__ void linearize(const float *x, const float *u, LaneModel<N, N> &md) const
    {
        float pre[kPre];
        prelinearize(x, u, pre);
        linearize_pre(pre, x, u, md);
    }
    __device__ void linearize_pre(const float *pre, const float *x, const float *u, LaneModel<N, N> &md) const
    {
        const float lam = pre[0];
        const float *grad = pre + 1;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            md.lx[i] = 2.0f * (x[i] - goal[i]);
            md.lu[i] = 0.0f;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float id = (i == j) ? 1.0f : 0.0f;
                md.fx(i, j) = id + u[i] * grad[j];
                md.fu(i, j) = lam * id;
                md.lxx(i, j) = 2.0f * id; md.luu(i, j) = 0.0f; md.lux(i, j) = 0.0f;
            }
        }
        md.l = final_cost(x);
    }
    __device__ float final_quad(const float *x, float *lx, Mat<N, N> &lxx) const
    {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            lx[i] = 2.0f * (x[i] - goal[i]);
#pragma unroll
            for (int j = 0; j < N; ++j) lxx(i, j) = (i == j) ? 2.0f : 0.0f;
        }
        return final_cost(x);
    }
};

// ---- box-QP (tfmpc/utils/optimization.py:6-101) in registers ----------------------------
template <int M>
__device__ __forceinline__ float qp_value(const Mat<M, M> &H, const float *q, const float *x)
{
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        float hx = 0.0f;
#pragma unroll
        for (int j = 0; j < M; ++j) hx = fmaf(H(i, j), x[j], hx);
        v += x[i] * (0.5f * hx + q[i]);
    }
    return v;
}

// Solves the free sub-system H_ff y = rhs_f (clamped dimensions become identity rows).
template <int M, int W>
__device__ __forceinline__ int solve_free(const Mat<M, M> &H, const bool *fre, const Mat<M, W> &rhs, Mat<M, W> &out)
{
    if constexpr (M == 2) {
        // two variables: the elimination written out (same pivots, same positivity test as gauss_jordan<.., false>;
        // a clamped variable is an identity row), two or three divisions instead of the generic in-register sweep
        const bool f0 = fre[0], f1 = fre[1];
        const float h00 = f0 ? H(0, 0) : 1.0f, h11 = f1 ? H(1, 1) : 1.0f;
        const float h01 = (f0 && f1) ? H(0, 1) : 0.0f, h10 = (f0 && f1) ? H(1, 0) : 0.0f;
        const float inv0 = lane_rcp(h00), p2 = fmaf(-h10, h01 * inv0, h11), inv1 = lane_rcp(p2);
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float r0 = f0 ? rhs(0, j) : 0.0f, r1 = f1 ? rhs(1, j) : 0.0f;
            const float y1 = fmaf(-h10, r0 * inv0, r1) * inv1;
            out(1, j) = y1;
            out(0, j) = fmaf(-h01 * inv0, y1, r0 * inv0);
        }
        return (!(h00 > 0.0f) || !(p2 > 0.0f)) ? 1 : 0;
    }
    Mat<M, M + W> aug;
#pragma unroll
    for (int i = 0; i < M; ++i) {
#pragma unroll
        for (int j = 0; j < M; ++j) aug(i, j) = (fre[i] && fre[j]) ? H(i, j) : ((i == j) ? 1.0f : 0.0f);
#pragma unroll
        for (int j = 0; j < W; ++j) aug(i, M + j) = fre[i] ? rhs(i, j) : 0.0f;
    }
    const int bad = small::gauss_jordan<M, W, false>(aug);
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j) out(i, j) = aug(i, M + j);
    return bad;
}

template <int M>
__device__ inline int boxqp_lane(const Mat<M, M> &H, const float *q, const float *lo, const float *hi, float *x, bool *fre)
{
    const float rtol = 1e-8f, step_dec = 0.6f, min_step = 1e-22f, armijo = 0.1f, eps = 1e-6f;
    if constexpr (M == 2) {
        // Two actions (BASELINE configs[3]; round 3): a strictly convex QP over a box has ONE Karush-Kuhn-Tucker point,
        // and with two variables there are nine places it can be -- the interior (both free), four edges (one variable
        // on a bound, the other free), four corners.  Each candidate is a handful of operations; the one that is
        // feasible with the right multiplier signs IS the minimiser the projected-Newton iteration of
        // optimization.py:24-99 converges to, and its free set is what the clamp rule (:121-127) gives there.  The
        // iteration itself (up to 100 passes of gradient, clamp set, elimination, Armijo loop with IEEE divisions) was
        // 130 k of the 259 k cycles of a cfg4 iLQR iteration.  Candidates are accepted only with a margin of `eps`
        // (= the rule's own tolerance, and the gradient tolerance 1e-6 of :13-17) on every comparison, and only if
        // exactly one qualifies; anything closer to a tie, and any H that is not positive definite, is left to the
        // iteration below (which also reports TFMPC_ST_NOT_PD as before).
        const float h00 = H(0, 0), h01 = H(0, 1), h10 = H(1, 0), h11 = H(1, 1);
        const float inv0 = lane_rcp(h00), p2 = fmaf(-h10, h01 * inv0, h11);   // the pivots of the first factorisation
        if (h00 > 0.0f && p2 > 0.0f && h11 > 0.0f) {
            const float inv1 = lane_rcp(h11);
            // a candidate: variable i on bound b_i (s_i = -1 lower, +1 upper) or free (s_i = 0); ok = feasible with
            // the right multiplier signs, every comparison with a margin of eps
            auto candidate = [&](int s0, int s1, float &a0, float &a1) {
                const float b0 = s0 > 0 ? hi[0] : lo[0], b1 = s1 > 0 ? hi[1] : lo[1];
                if (s0 == 0 && s1 == 0) {                                            // H x = -q
                    a1 = (h10 * (q[0] * inv0) - q[1]) * lane_rcp(p2);
                    a0 = -(q[0] + h01 * a1) * inv0;
                    return a0 > lo[0] + eps && a0 < hi[0] - eps && a1 > lo[1] + eps && a1 < hi[1] - eps;
                }
                if (s0 != 0 && s1 == 0) {
                    a0 = b0;
                    a1 = -(q[1] + h10 * b0) * inv1;
                    const float g0 = fmaf(h00, b0, fmaf(h01, a1, q[0]));
                    return a1 > lo[1] + eps && a1 < hi[1] - eps && (s0 > 0 ? g0 < -eps : g0 > eps);
                }
                if (s0 == 0 && s1 != 0) {
                    a1 = b1;
                    a0 = -(q[0] + h01 * b1) * inv0;
                    const float g1 = fmaf(h10, a0, fmaf(h11, b1, q[1]));
                    return a0 > lo[0] + eps && a0 < hi[0] - eps && (s1 > 0 ? g1 < -eps : g1 > eps);
                }
                a0 = b0; a1 = b1;
                const float g0 = fmaf(h00, b0, fmaf(h01, b1, q[0])), g1 = fmaf(h10, b0, fmaf(h11, b1, q[1]));
                return (s0 > 0 ? g0 < -eps : g0 > eps) && (s1 > 0 ? g1 < -eps : g1 > eps);
            };
            float a0, a1;
            // the usual case first: the unconstrained minimiser, else the active set its violations suggest -- a
            // candidate that passes IS the solution (the KKT point is unique), so nothing else needs looking at
            if (candidate(0, 0, a0, a1)) { x[0] = a0; x[1] = a1; fre[0] = fre[1] = true; return 0; }
            {
                const int s0 = a0 <= lo[0] + eps ? -1 : (a0 >= hi[0] - eps ? 1 : 0);
                const int s1 = a1 <= lo[1] + eps ? -1 : (a1 >= hi[1] - eps ? 1 : 0);
                float c0, c1;
                if ((s0 != 0 || s1 != 0) && candidate(s0, s1, c0, c1)) {
                    x[0] = c0; x[1] = c1; fre[0] = s0 == 0; fre[1] = s1 == 0;
                    return 0;
                }
            }
            // rare: all nine, accepted only if exactly one qualifies
            float cx0 = 0.0f, cx1 = 0.0f;
            int cs0 = 0, cs1 = 0, hits = 0;
#pragma unroll
            for (int s0 = -1; s0 <= 1; ++s0)
#pragma unroll
                for (int s1 = -1; s1 <= 1; ++s1) {
                    float c0, c1;
                    if (candidate(s0, s1, c0, c1)) { cx0 = c0; cx1 = c1; cs0 = s0; cs1 = s1; hits += 1; }
                }
            if (hits == 1) {
                x[0] = cx0; x[1] = cx1;
                fre[0] = cs0 == 0; fre[1] = cs1 == 0;
                return 0;
            }
        }
    }
    float value = qp_value<M>(H, q, x), old_value = value;
#pragma unroll
    for (int i = 0; i < M; ++i) fre[i] = true;
    for (int it = 0; it < 100; ++it) {
        if (it > 0 && (old_value - value) < rtol * fabsf(old_value)) return 0;
        old_value = value;
        float g[M];
        int n_free = 0;
        float gn = 0.0f;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            float gi = q[i];
#pragma unroll
            for (int j = 0; j < M; ++j) gi = fmaf(H(i, j), x[j], gi);
            g[i] = gi;
            const bool clamped = (fabsf(x[i] - lo[i]) < eps && gi > 0.0f) || (fabsf(hi[i] - x[i]) < eps && gi < 0.0f);
            fre[i] = !clamped;
            if (!clamped) { n_free += 1; gn = fmaf(gi, gi, gn); }
        }
        Mat<M, 1> gc, sol;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            float s = q[i];
#pragma unroll
            for (int j = 0; j < M; ++j) s = fmaf(H(i, j), fre[j] ? 0.0f : x[j], s);
            gc(i, 0) = s;
        }
        if (solve_free<M, 1>(H, fre, gc, sol)) return (it == 0) ? TFMPC_ST_NOT_PD : 0;
        if (n_free == 0) return 0;
        if (sqrtf(gn) < eps) return 0;
        float srch[M], sdotg = 0.0f;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            srch[i] = fre[i] ? (-sol(i, 0) - x[i]) : 0.0f;
            sdotg = fmaf(srch[i], g[i], sdotg);
        }
        if (sdotg >= 0.0f) return 0;
        float step = 1.0f, xc[M], vc;
        for (;;) {
#pragma unroll
            for (int i = 0; i < M; ++i) xc[i] = fminf(fmaxf(fmaf(step, srch[i], x[i]), lo[i]), hi[i]);
            vc = qp_value<M>(H, q, xc);
            if (!((vc - old_value) / (step * sdotg) < armijo)) break;
            step *= step_dec;
            if (step < min_step) {
#pragma unroll
                for (int i = 0; i < M; ++i) xc[i] = fminf(fmaxf(fmaf(step, srch[i], x[i]), lo[i]), hi[i]);
                vc = qp_value<M>(H, q, xc);
                break;
            }
        }
#pragma unroll
        for (int i = 0; i < M; ++i) x[i] = xc[i];
        value = vc;
    }
    return TFMPC_ST_QP_MAXITER;
}

// ---- where a lane keeps its nominal trajectory and gains --------------------------------
// GlobalStore: the instance's HBM slabs (any horizon).  LdsStore: the wave's LDS, laid out
// [slot][lane] so the 64 lanes of a wave hit 64 different banks; removes the dependent HBM
// round trip from every step of the backward sweep and of the up-to-11 rollouts.
template <int N, int M>
struct GlobalStore {
    float *xh, *uh, *Kg, *kg;
    __device__ float x(int t, int i) const { return xh[(size_t)t * N + i]; }
    __device__ float u(int t, int a) const { return uh[(size_t)t * M + a]; }
    __device__ float K(int t, int a, int j) const { return Kg[(size_t)t * M * N + a * N + j]; }
    __device__ float k(int t, int a) const { return kg[(size_t)t * M + a]; }
    __device__ void set_x(int t, int i, float v) { xh[(size_t)t * N + i] = v; }
    __device__ void set_u(int t, int a, float v) { uh[(size_t)t * M + a] = v; }
    __device__ void set_K(int t, int a, int j, float v) { Kg[(size_t)t * M * N + a * N + j] = v; }
    __device__ void set_k(int t, int a, float v) { kg[(size_t)t * M + a] = v; }
};

template <int N, int M>
struct LdsStore {
    static constexpr int kPerStep = N + M + M * N + M;      // x, u, K, k of one timestep
    float *base;                                            // wave's LDS + lane
    __device__ float &at(int t, int off) const { return base[(t * kPerStep + off) * 64]; }
    __device__ float x(int t, int i) const { return at(t, i); }
    __device__ float u(int t, int a) const { return at(t, N + a); }
    __device__ float K(int t, int a, int j) const { return at(t, N + M + a * N + j); }
    __device__ float k(int t, int a) const { return at(t, N + M + M * N + a); }
    __device__ void set_x(int t, int i, float v) { at(t, i) = v; }
    __device__ void set_u(int t, int a, float v) { at(t, N + a) = v; }
    __device__ void set_K(int t, int a, int j, float v) { at(t, N + M + a * N + j) = v; }
    __device__ void set_k(int t, int a, float v) { at(t, N + M + M * N + a) = v; }
    static size_t bytes(int T) { return (size_t)(T + 1) * kPerStep * 64 * sizeof(float); }
};

// -DTFMPC_PHASE_PROBE (tools/probes/lane_probe.py): every lane accumulates s_memtime deltas per phase of an iteration;
// the group kernel leaves those of instance 0 in costs[0..7] (as floats) instead of the stage costs.
#ifdef TFMPC_PHASE_PROBE
struct LaneProbe { long long acc[8]; long long last; };
__device__ LaneProbe g_probe_dummy;
#define TFMPC_PROBE_ARG , LaneProbe &pr
#define TFMPC_PROBE_PASS , pr
#define TFMPC_PROBE_START() (pr.last = __builtin_amdgcn_s_memtime())
#define TFMPC_PROBE(i) do { const long long now_ = __builtin_amdgcn_s_memtime(); pr.acc[i] += now_ - pr.last; pr.last = now_; } while (0)
#else
#define TFMPC_PROBE_ARG
#define TFMPC_PROBE_PASS
#define TFMPC_PROBE_START() ((void)0)
#define TFMPC_PROBE(i) ((void)0)
#endif

// ---- the solve ------------------------------------------------------------------------
struct LaneBackward { float J, dV1, dV2, g_norm; int failed, flags; };

struct SolveArgsLane {
    int B, T;
    const float *x0, *u_init;
    float *states, *actions, *costs;
    int32_t *iterations, *status;
    float *wsK, *wsk, *wsx, *wsu, *wsc;
    float *scratch;          // group kernel only: one block of candidate trajectories per wavefront
    int stored;              // group kernel only: how many step sizes of a group keep their candidate (ScratchSink)
    int *queue;              // group kernel only: the next instance nobody has taken yet (zeroed by the launcher)
    TraceArgs trace;         // group kernel only: the optional decision trace
};
// (TFMPC_GROUP_STORED = 1 .. 4, tests and A/B timing: fewer columns of the block in use -- 1 = almost every backtracking pass replays; what the
// default is about: ScratchSink)
constexpr int kStoredCandidates = 4;
inline int group_stored_candidates(int requested) { return requested < 1 || requested > kStoredCandidates ? kStoredCandidates : requested; }

template <int KIND, int N, int M, class Store, bool PRE = false>
__device__ inline LaneBackward backward_lane(const LaneEnv<KIND, N, M> &env, int T, float mu, bool bounded,
                                             const float *low, const float *high, Store &st TFMPC_PROBE_ARG)
{
    LaneBackward r{0.0f, 0.0f, 0.0f, 0.0f, 0, 0};
    float Vx[N];
    Mat<N, N> Vxx;
    {
        float xT[N];
#pragma unroll
        for (int i = 0; i < N; ++i) xT[i] = st.x(T, i);
        r.J = env.final_quad(xT, Vx, Vxx);                                  // ilqr.py:101-104
    }
    float gsum = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
        float x[N], u[M];
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = st.x(t, i);
#pragma unroll
        for (int a = 0; a < M; ++a) u[a] = st.u(t, a);
        LaneModel<N, M> md;
        TFMPC_PROBE_START();
        if constexpr (PRE && LaneEnv<KIND, N, M>::kPre > 0) {
            float pre[LaneEnv<KIND, N, M>::kPre];
#pragma unroll
            for (int j = 0; j < LaneEnv<KIND, N, M>::kPre; ++j) pre[j] = st.pre(t, j);
            env.linearize_pre(pre, x, u, md);
        } else {
            env.linearize(x, u, md);
        }
        TFMPC_PROBE(0);
        float Qx[N], Qu[M];
#pragma unroll
        for (int i = 0; i < N; ++i) {                                       // :122
            float s = md.lx[i];
#pragma unroll
            for (int k = 0; k < N; ++k) s = fmaf(md.fx(k, i), Vx[k], s);
            Qx[i] = s;
        }
#pragma unroll
        for (int a = 0; a < M; ++a) {                                       // :123
            float s = md.lu[a];
#pragma unroll
            for (int k = 0; k < N; ++k) s = fmaf(md.fu(k, a), Vx[k], s);
            Qu[a] = s;
        }
        const Mat<N, N> W1 = small::mul_tn<N, N, N>(md.fx, Vxx);            // :125
        const Mat<M, N> W2 = small::mul_tn<N, M, N>(md.fu, Vxx);            // :126
        bool vxx_nonzero = false;
#pragma unroll
        for (int i = 0; i < N * N; ++i) vxx_nonzero = vxx_nonzero || (Vxx.a[i] != 0.0f);
        Mat<N, N> Qxx;
        Mat<M, M> Quu, Quur;
        Mat<M, N> Qux, Quxr;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {                                   // :129
                float s = md.lxx(i, j);
#pragma unroll
                for (int k = 0; k < N; ++k) s = fmaf(W1(i, k), md.fx(k, j), s);
                Qxx(i, j) = s;
            }
#pragma unroll
        for (int a = 0; a < M; ++a) {
#pragma unroll
            for (int c = 0; c < M; ++c) {                                   // :130, :133
                float s = md.luu(a, c), sr = md.luu(a, c);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    s = fmaf(W2(a, k), md.fu(k, c), s);
                    sr = fmaf(fmaf(mu, md.fu(k, a), W2(a, k)), md.fu(k, c), sr);
                }
                Quu(a, c) = s; Quur(a, c) = sr;
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {                                   // :131, :134
                float s = md.lux(a, j), sr = md.lux(a, j);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    s = fmaf(W2(a, k), md.fx(k, j), s);
                    sr = fmaf(fmaf(mu, md.fu(k, a), W2(a, k)), md.fx(k, j), sr);
                }
                Qux(a, j) = s; Quxr(a, j) = sr;
            }
        }
        Mat<M, N> K;
        float kk[M];
        TFMPC_PROBE(1);
        if (!bounded) {                                                     // :357-362
            Mat<M, M + 1 + N> aug;
#pragma unroll
            for (int a = 0; a < M; ++a) {
#pragma unroll
                for (int c = 0; c < M; ++c) aug(a, c) = Quur(a, c);
                aug(a, M) = Qu[a];
#pragma unroll
                for (int j = 0; j < N; ++j) aug(a, M + 1 + j) = Quxr(a, j);
            }
            if (small::gauss_jordan<M, 1 + N, false>(aug)) { r.failed = 1; return r; }
#pragma unroll
            for (int a = 0; a < M; ++a) {
                kk[a] = -aug(a, M);
#pragma unroll
                for (int j = 0; j < N; ++j) K(a, j) = -aug(a, M + 1 + j);
            }
        } else if (vxx_nonzero) {                                           // :364-387
            float lo[M], hi[M];
            bool fre[M];
#pragma unroll
            for (int a = 0; a < M; ++a) {
                lo[a] = low[a] - u[a]; hi[a] = high[a] - u[a];
                kk[a] = (lo[a] + hi[a]) / 2;
            }
            const int rc = boxqp_lane<M>(Quur, Qu, lo, hi, kk, fre);
            TFMPC_PROBE(2);
            if (rc == TFMPC_ST_NOT_PD) { r.failed = 1; return r; }
            r.flags |= rc;
            Mat<M, N> sol;
            if (solve_free<M, N>(Quur, fre, Quxr, sol)) { r.failed = 1; return r; }
#pragma unroll
            for (int a = 0; a < M; ++a)
#pragma unroll
                for (int j = 0; j < N; ++j) K(a, j) = fre[a] ? -sol(a, j) : 0.0f;
        } else {                                                            // :140-141
#pragma unroll
            for (int a = 0; a < M; ++a) {
                kk[a] = (Qu[a] >= 0.0f) ? (low[a] - u[a]) : (high[a] - u[a]);
#pragma unroll
                for (int j = 0; j < N; ++j) K(a, j) = 0.0f;
            }
        }
        TFMPC_PROBE(3);
        Mat<N, M> KtQ;                                                      // :147
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int c = 0; c < M; ++c) {
                float s = 0.0f;
#pragma unroll
                for (int a = 0; a < M; ++a) s = fmaf(K(a, i), Quu(a, c), s);
                KtQ(i, c) = s;
            }
        Mat<N, N> Vn;
#pragma unroll
        for (int i = 0; i < N; ++i) {                                       // :149-161
            float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
            for (int a = 0; a < M; ++a) {
                s1 = fmaf(Qux(a, i), kk[a], s1);
                s2 = fmaf(K(a, i), Qu[a], s2);
                s3 = fmaf(KtQ(i, a), kk[a], s3);
            }
            Vx[i] = Qx[i] + s1 + s2 + s3;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
#pragma unroll
                for (int a = 0; a < M; ++a) {
                    t1 = fmaf(Qux(a, i), K(a, j), t1);
                    t2 = fmaf(K(a, i), Qux(a, j), t2);
                    t3 = fmaf(KtQ(i, a), K(a, j), t3);
                }
                Vn(i, j) = Qxx(i, j) + t1 + t2 + t3;
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) Vxx(i, j) = 0.5f * (Vn(i, j) + Vn(j, i));   // :162
        float p1 = 0.0f, p2 = 0.0f, gmax = 0.0f;                            // :164-167, :243
#pragma unroll
        for (int a = 0; a < M; ++a) {
            float quk = 0.0f;
#pragma unroll
            for (int c = 0; c < M; ++c) quk = fmaf(Quu(a, c), kk[c], quk);
            p1 = fmaf(kk[a], Qu[a], p1);
            p2 = fmaf(kk[a], quk, p2);
            gmax = fmaxf(gmax, fabsf(kk[a]) / (fabsf(u[a]) + 1.0f));
            st.set_k(t, a, kk[a]);
#pragma unroll
            for (int j = 0; j < N; ++j) st.set_K(t, a, j, K(a, j));
        }
        r.J += md.l;
        r.dV1 += p1;
        r.dV2 += 0.5f * p2;
        gsum += gmax;
        TFMPC_PROBE(4);
    }
    r.g_norm = T > 0 ? gsum / (float)T : 0.0f;
    return r;
}

// Where a rollout's trajectory goes: HBM slabs, nowhere (speculative rollouts only need J and
// the residual), or the group's second nominal buffer.
template <int N, int M>
struct GlobalSink {
    float *xs, *us, *cs;
    __device__ void x(int t, int i, float v) const { xs[(size_t)t * N + i] = v; }
    __device__ void u(int t, int a, float v) const { us[(size_t)t * M + a] = v; }
    __device__ void c(int t, float v) const { cs[t] = v; }
};
struct NullSink {
    __device__ void x(int, int, float) const {}
    __device__ void u(int, int, float) const {}
    __device__ void c(int, float) const {}
};

template <int KIND, int N, int M, class Store, class Sink>
__device__ inline void forward_lane(const LaneEnv<KIND, N, M> &env, int T, float alpha, const float *low,
                                    const float *high, const Store &st, const Sink &sink, float &J_out,
                                    float &res_out)
{
    float x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] = st.x(0, i); sink.x(0, i, x[i]); }
    float J = 0.0f, resid = 0.0f;
    for (int t = 0; t < T; ++t) {                                           // ilqr.py:192-206
        float u[M], xn[N];
#pragma unroll
        for (int a = 0; a < M; ++a) {
            float du = alpha * st.k(t, a);
#pragma unroll
            for (int j = 0; j < N; ++j) du = fmaf(st.K(t, a, j), x[j] - st.x(t, j), du);
            u[a] = fminf(fmaxf(st.u(t, a) + du, low[a]), high[a]);
            sink.u(t, a, u[a]);
            resid = fmaxf(resid, fabsf(du));
        }
        const float c = env.cost(x, u);
        env.transition(x, u, xn);
        J += c;
        sink.c(t, c);
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] = xn[i]; sink.x(t + 1, i, xn[i]); }
    }
    const float fc = env.final_cost(x);
    sink.c(T, fc);
    J_out = J + fc;
    res_out = resid;
}

template <int KIND, int N, int M, bool USE_LDS>
__global__ __launch_bounds__(64) void ilqr_lane_solve_kernel(TfmpcEnv genv, TfmpcIlqrConfig cfg, SolveArgsLane a)
{
    extern __shared__ float lane_lds[];
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const int T = a.T;
    LaneEnv<KIND, N, M> env;
    env.load(genv, b);
    float low[M], high[M];
#pragma unroll
    for (int i = 0; i < M; ++i) { low[i] = genv.low[i]; high[i] = genv.high[i]; }
    const bool bounded = genv.bounded != 0;

    float *xout = a.states + (size_t)b * (T + 1) * N, *uout = a.actions + (size_t)b * T * M,
          *chat = a.costs + (size_t)b * (T + 1);
    float *xc = a.wsx + (size_t)b * (T + 1) * N, *uc = a.wsu + (size_t)b * T * M, *cc = a.wsc + (size_t)b * (T + 1);
    using Store = typename std::conditional<USE_LDS, LdsStore<N, M>, GlobalStore<N, M>>::type;
    Store st;
    if constexpr (USE_LDS) st.base = lane_lds + threadIdx.x;
    else { st.xh = xout; st.uh = uout; st.Kg = a.wsK + (size_t)b * T * M * N; st.kg = a.wsk + (size_t)b * T * M; }

    {   // start (ilqr.py:218)
        float x[N], xn[N], u[M];
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] = a.x0[(size_t)b * N + i]; st.set_x(0, i, x[i]); }
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int i = 0; i < M; ++i) { u[i] = a.u_init[((size_t)b * T + t) * M + i]; st.set_u(t, i, u[i]); }
            chat[t] = env.cost(x, u);
            env.transition(x, u, xn);
#pragma unroll
            for (int i = 0; i < N; ++i) { x[i] = xn[i]; st.set_x(t + 1, i, xn[i]); }
        }
        chat[T] = env.final_cost(x);
    }

    float mu = 0.0f, delta = 1.0f;
    int status = 0, attempts = 0, iteration = 0;
    bool converged = false, give_up = false;
#ifdef TFMPC_PHASE_PROBE
    LaneProbe pr{};
#endif
    for (iteration = 0; iteration < cfg.max_iterations; ++iteration) {
        for (;;) {
            float mu_l = mu, delta_l = delta;
            LaneBackward r;
            for (int retry = 0;; ++retry) {                                  // :285-315
                r = backward_lane<KIND, N, M>(env, T, mu_l, bounded, low, high, st TFMPC_PROBE_PASS);
                status |= r.flags;
                if (!r.failed) break;
                status |= TFMPC_ST_NOT_PD;
                delta_l = fmaxf(cfg.delta_0, delta_l * cfg.delta_0);
                mu_l = fmaxf(cfg.mu_min, mu_l * delta_l);
                if (retry >= 40) { give_up = true; break; }
            }
            if (give_up) break;
            if (r.g_norm < cfg.atol) { converged = true; break; }            // :243-248
            bool accept = false;
            float residual = 0.0f;
            for (int ai = 0; ai < cfg.n_alphas; ++ai) {                      // :317-355
                const float alpha = cfg.alphas[ai];
                float J;
                forward_lane<KIND, N, M>(env, T, alpha, low, high, st, GlobalSink<N, M>{xc, uc, cc}, J, residual);
                const float delta_J = -alpha * (r.dV1 + alpha * r.dV2);
                const float dcost = r.J - J;
                const float z = (delta_J > 0.0f) ? dcost / delta_J : sgnf(dcost);
                if (z >= cfg.c1) { accept = true; break; }
            }
            const bool small_step = residual < cfg.atol;                    // :253-257
            if (small_step || accept) {
                for (int t = 0; t <= T; ++t) {
#pragma unroll
                    for (int i = 0; i < N; ++i) st.set_x(t, i, xc[(size_t)t * N + i]);
                    chat[t] = cc[t];
                }
                for (int t = 0; t < T; ++t) {
#pragma unroll
                    for (int i = 0; i < M; ++i) st.set_u(t, i, uc[(size_t)t * M + i]);
                }
            }
            if (small_step) { converged = true; break; }
            if (accept) {                                                    // :259-266
                delta = fminf(1.0f / cfg.delta_0, delta / cfg.delta_0);
                mu = (mu * delta > cfg.mu_min) ? mu * delta : 0.0f;
                break;
            }
            delta = fmaxf(cfg.delta_0, delta * cfg.delta_0);                 // :267-270
            mu = fmaxf(cfg.mu_min, mu * delta);
            if (++attempts >= cfg.max_attempts || !(mu < 1e30f)) { give_up = true; break; }
        }
        if (converged || give_up) break;
    }
    if (give_up) status |= TFMPC_ST_MAX_ATTEMPTS;
    if (iteration >= cfg.max_iterations) iteration = cfg.max_iterations - 1;
    if constexpr (USE_LDS) {        // the nominal trajectory leaves LDS once, at the end
        for (int t = 0; t <= T; ++t) {
#pragma unroll
            for (int i = 0; i < N; ++i) xout[(size_t)t * N + i] = st.x(t, i);
        }
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int i = 0; i < M; ++i) uout[(size_t)t * M + i] = st.u(t, i);
        }
    }
    const float cT = chat[T];
    if (!(cT == cT)) status |= TFMPC_ST_NAN;
    a.iterations[b] = iteration;
    a.status[b] = status;
}

// Batches up to this size run one group per wavefront in a single launch (B waves fit the chip at <= 2 per SIMD).
constexpr int kOneGroupMaxBatch = 2048;
// ---- group-per-instance variant: speculative PARALLEL line search ---------------------------
// 16 lanes own one instance.  What is sequential in the reference (the backward sweep with its box-QPs) runs
// redundantly on all 16 lanes -- same data, same control flow, so no broadcast is needed -- and everything that is
// NOT sequential is spread over the lanes:
//   * the x-only part of the linearisation (LaneEnv::prelinearize: the deceleration factor and its gradient, a third of
//     an iteration when it sat inside the sweep) is evaluated for all timesteps at once, one timestep per lane, into LDS;
//   * the 11 line-search rollouts of ilqr.py:322 run AT ONCE, one step size per lane, each lane leaving its candidate
//     trajectory in its own column of a per-wave HBM scratch (coalesced fire-and-forget stores); a ballot picks the
//     first accepted step size -- exactly the reference's "first alpha that passes" -- and the group copies that column
//     into its nominal buffer (no second rollout).
// Arithmetic per instance is that of ilqr_lane_solve_kernel, bit for bit (tested).
//
// GROUPS instances share a wavefront: 4 for large batches (the chip is full and latency-bound; the groups of a wave
// pay for each other's divergent box-QP trip counts), 1 while the chip has room for a wavefront per instance (0.55x the
// latency per iteration).  The launch lasts as long as its slowest instance (cfg4: median 8 iterations, p99 20, max 87).
// Tried and measured worse (round 2, tools/cfg4_sustained.py): stopping the main launch after 8 .. 32 iterations and
// resuming the parked stragglers in a tail launch with a wavefront each -- a straggler that stays in the main launch
// already runs alone in its wave once its neighbours have converged, so the cut only adds a launch and a reload
// (single batch 13.1 -> 15.1 .. 16.5 ms; 8 batches in flight 4.2 -> 5.1 ms per batch).
template <int N, int M, int GROUPS, int PRE>
struct GroupStore {
    float *base;                 // wave's LDS + group index; slot stride = GROUPS
    int T;

    __device__ int uoff() const { return (T + 1) * N; }
    __device__ int Koff() const { return (T + 1) * N + T * M; }
    __device__ int koff() const { return Koff() + T * M * N; }
    __device__ int poff() const { return koff() + T * M; }
    __device__ float &at(int slot) const { return base[slot * GROUPS]; }
    __device__ float x(int t, int i) const { return at(t * N + i); }
    __device__ float u(int t, int a) const { return at(uoff() + t * M + a); }
    __device__ float K(int t, int a, int j) const { return at(Koff() + t * M * N + a * N + j); }
    __device__ float k(int t, int a) const { return at(koff() + t * M + a); }
    __device__ float pre(int t, int j) const { return at(poff() + t * PRE + j); }
    __device__ void set_x(int t, int i, float v) { at(t * N + i) = v; }
    __device__ void set_u(int t, int a, float v) { at(uoff() + t * M + a) = v; }
    __device__ void set_K(int t, int a, int j, float v) { at(Koff() + t * M * N + a * N + j) = v; }
    __device__ void set_k(int t, int a, float v) { at(koff() + t * M + a) = v; }
    __device__ void set_pre(int t, int j, float v) { at(poff() + t * PRE + j) = v; }
    static size_t bytes(int T) { return (size_t)((T + 1) * N + T * M + T * M * N + T * M + T * PRE) * GROUPS * sizeof(float); }
};

// Candidate trajectory of one lane's speculative rollout: rows [x_1 .. x_T | u_0 .. u_{T-1} | c_0 .. c_T] of the
// group's scratch, one column per STORED step size (x_0 never changes).
// Round 6: only the first `stored` step sizes of a group keep their candidates (default kStoredCandidates = 4: 99.8 % of cfg4's passes adopt
// index 0 .. 3 -- 67 439 / 40 349 / 14 888 / 3 400 of 126 321, and the slowest instances, which set the launch time, index 3 --,
// tools/probes/r6_cfg4_alpha_hist.py), in a block of the group's own: [row][4] floats, 16 bytes a row.  Every lane used to write its column
// of a [row][64 lanes] block -- 64 KB per wave and pass, of which the adoption read one column at 64-byte sector granularity: 4.35 GB of HBM traffic per
// launch of 16 384 instances against 0.26 GB algorithmic (the round-5 verdict's item 8).  A pass that adopts a step size beyond the stored ones (0.2 %)
// rolls THAT step size out once more on the whole group into column 0 -- the same function on the same inputs: the same bits -- and adopts it from there.
template <int N, int M>
struct ScratchSink {
    static constexpr int kStride = 4;     // floats between rows = columns of the block (kStoredCandidates; a constant: the row index times a
                                          // run-time stride is an integer multiply per store, five per lane and time step -- + 8 % on the user-env form)
    float *col;                  // group's scratch block + this lane's column
    int T;
    bool live;                   // only the stored step sizes keep their candidate
    __host__ __device__ static int rows(int T) { return T * N + T * M + T + 1; }
    __device__ void x(int t, int i, float v) const { if (live && t > 0) col[((t - 1) * N + i) * kStride] = v; }
    __device__ void u(int t, int a, float v) const { if (live) col[(T * N + t * M + a) * kStride] = v; }
    __device__ void c(int t, float v) const { if (live) col[(T * N + T * M + t) * kStride] = v; }
};

// Round 4: PERSISTENT groups with an instance QUEUE.  A launch used to give every group exactly one instance, so a wave lasted as
// long as the slowest of its four instances (cfg4: median 8 iterations, p99 20, max 87) and a batch as long as its slowest wave;
// the sustained rate needed eight batches in flight on eight host streams.  Now the grid is what the chip holds at once and a
// group whose instance has finished takes the next one from an atomic counter (`a.queue`, zeroed by the launcher): ONE launch of
// any number of instances keeps every group busy until the queue is empty.  For that the solve is a flat state machine -- one
// trip of the loop below = (take an instance and roll out its start) | (one backward pass + line search of the group's
// instance) -- because lanes that leave a NESTED loop early wait at its exit for the rest of the wave: with the reference's loop
// nest (ilqr.py:227, :238) a finished group would have waited for its neighbours anyway, and a group whose step was accepted
// waited for a neighbour's rejected passes.  Every instance still runs exactly the reference's sequence of passes with the same
// arithmetic: outputs are bit-identical to the one-instance-per-group launch (tested against the per-lane kernel as before).
// Waves per SIMD the register budget is sized for.  3 (round 4): the flat state machine needs 148 VGPRs; sized for 4 (128 VGPRs) it
// spilled 19 of them inside the passes.  Measured (tools/probes/r4_cfg4_eu.sh, three alternating runs): single batch of 16 384
// 8.7 - 9.0 -> 7.9 - 8.5 ms, one launch of 131 072 instances 20.3 - 20.8 -> 20.0 - 20.6 ms.
#ifndef TFMPC_GROUP_LANE_EU
#define TFMPC_GROUP_LANE_EU 3
#endif
// Round 6: EU is a template parameter, because a USER env's functions decide how many registers the kernel needs and an instantiation that SPILLS
// vector registers must not be launched: this kernel keeps ~120 scalars in lanes of vector registers (SGPR spills), leaves the wave with part of
// its lanes retired (GROUPS = 1) or diverged, and with vector-register spills on top the emitted code loses scalars -- measured: wrong status words,
// a hang, a memory fault (an LQ env as 2 x 2 user source, 37 spilled registers at EU = 3; clean at EU = 1).  The launchers therefore ask the runtime
// for the instantiation's private segment and register count and take one that spills nothing (user_env_kernels.hip.in: EU = 3, then 1, else
// another kernel); tests/test_lane_group_private_segment_cpu.py holds the library's own instantiations to that.
template <int KIND, int N, int M, int GROUPS, int EU = TFMPC_GROUP_LANE_EU>
__global__ __launch_bounds__(64, EU) void ilqr_group_solve_kernel(TfmpcEnv genv, TfmpcIlqrConfig cfg, SolveArgsLane a)
{
    extern __shared__ float lane_lds[];
    constexpr int G = 16;
    constexpr int PRE = LaneEnv<KIND, N, M>::kPre;
    using Store = GroupStore<N, M, GROUPS, PRE>;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    if (grp >= GROUPS) return;                      // one group per wave: lanes 16..63 idle
    const int T = a.T;
    float low[M], high[M];
#pragma unroll
    for (int i = 0; i < M; ++i) { low[i] = genv.low[i]; high[i] = genv.high[i]; }
    const bool bounded = genv.bounded != 0;
    const bool leader = gl == 0;
    const int my_alpha_idx = (gl < cfg.n_alphas) ? gl : cfg.n_alphas - 1;
    const float my_alpha = cfg.alphas[my_alpha_idx];
    // this group's block of the wave's scratch: [row][4] (see ScratchSink); a.stored: how many of the four columns are in use (TFMPC_GROUP_STORED)
    const int stored = a.stored;
    float *scratch = a.scratch + ((size_t)blockIdx.x * GROUPS + grp) * ScratchSink<N, M>::rows(T) * G;
    const ScratchSink<N, M> sink{scratch + (gl < stored ? gl : 0), T, gl < stored && gl < cfg.n_alphas};
    Store st{lane_lds + grp, T};

    // ---- per-group state of the machine --------------------------------------------------------------------------------
    LaneEnv<KIND, N, M> env;
    int b = 0;
    bool have = false, exhausted = false, need_pre = false;
    float mu = 0.0f, delta = 1.0f;
    int status = 0, attempts = 0, iteration = 0;
#ifdef TFMPC_PHASE_PROBE
    LaneProbe pr{};
#endif
    for (;;) {
        if (!have && !exhausted) {
            // ---- take the next instance (the leader asks, the group hears) and roll out its start (ilqr.py:218) --------
            int next = 0;
            if (leader) next = atomicAdd(a.queue, 1);
            next = __shfl(next, grp * G, 64);
            if (next >= a.B) {
                exhausted = true;
            } else {
                b = next;
                have = true;
                env.load(genv, b);
                mu = 0.0f; delta = 1.0f;                                         // :215-216
                status = 0; attempts = 0; iteration = 0;
                need_pre = true;
                float *chat0 = a.costs + (size_t)b * (T + 1);
                float x[N], xn[N], u[M];
#pragma unroll
                for (int i = 0; i < N; ++i) { x[i] = a.x0[(size_t)b * N + i]; st.set_x(0, i, x[i]); }
                for (int t = 0; t < T; ++t) {
#pragma unroll
                    for (int i = 0; i < M; ++i) { u[i] = a.u_init[((size_t)b * T + t) * M + i]; st.set_u(t, i, u[i]); }
                    const float c = env.cost(x, u);
                    if (leader) chat0[t] = c;
                    env.transition(x, u, xn);
#pragma unroll
                    for (int i = 0; i < N; ++i) { x[i] = xn[i]; st.set_x(t + 1, i, xn[i]); }
                }
                const float fc = env.final_cost(x);
                if (leader) chat0[T] = fc;
            }
        }
        if (!__any(have)) break;                    // every group of the wave has found the queue empty
        if (have) {
            float *xout = a.states + (size_t)b * (T + 1) * N, *uout = a.actions + (size_t)b * T * M,
                  *chat = a.costs + (size_t)b * (T + 1);
            bool finished = false, give_up = false;
            if constexpr (PRE > 0) {
                if (need_pre) {
                    // x-only part of the linearisation of every timestep, one timestep per lane (the nominal trajectory is
                    // fixed until a candidate is adopted, also across the regularisation retries)
                    TFMPC_PROBE_START();
                    for (int t = gl; t < T; t += G) {
                        float x[N], u[M], pre[PRE];
#pragma unroll
                        for (int i = 0; i < N; ++i) x[i] = st.x(t, i);
#pragma unroll
                        for (int i = 0; i < M; ++i) u[i] = st.u(t, i);
                        env.prelinearize(x, u, pre);
#pragma unroll
                        for (int j = 0; j < PRE; ++j) st.set_pre(t, j, pre[j]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    TFMPC_PROBE(7);
                }
            }
            need_pre = false;
            // ---- one pass through the body of ilqr.py:238-270 ------------------------------------------------------------
            float mu_l = mu, delta_l = delta;
            LaneBackward r;
            int level = 0;                                                       // local bumps before the sweep factorised (trace column)
            for (int retry = 0;; ++retry) {                                      // :285-315
                r = backward_lane<KIND, N, M, Store, true>(env, T, mu_l, bounded, low, high, st TFMPC_PROBE_PASS);
                status |= r.flags;
                if (!r.failed) break;
                status |= TFMPC_ST_NOT_PD;
                delta_l = fmaxf(cfg.delta_0, delta_l * cfg.delta_0);
                mu_l = fmaxf(cfg.mu_min, mu_l * delta_l);
                if (retry >= 40) { give_up = true; break; }
                ++level;
            }
            if (give_up) {
                finished = true;
            } else if (r.g_norm < cfg.atol) {                                    // :243-248
                if (leader) trace_write(a.trace, b, iteration + attempts, iteration, mu, delta, r.J, r.g_norm, -1, 0.0f, 0.0f, -1, -1.0f, level);
                finished = true;
            } else {
                // all step sizes at once, one per lane (ilqr.py:322-353), candidates into the scratch columns.  ONE call site of the rollout: a pass
                // that adopts a step size whose candidate was not kept (ScratchSink) takes a second trip through it -- the whole group on that step
                // size, lane 0 writing column 0 (a second inlined copy of the rollout cost the user-env instantiation of this kernel 10 %)
                float J = 0.0f, residual = 0.0f, res_chosen = 0.0f;
                bool accept = false, small_step = false;
                int chosen = 0, column = 0;
                float alpha_roll = my_alpha;
                ScratchSink<N, M> sink_roll = sink;
                TFMPC_PROBE_START();
                for (int trip = 0; trip < 2; ++trip) {
                    float J_trip, residual_trip;
                    forward_lane<KIND, N, M>(env, T, alpha_roll, low, high, st, sink_roll, J_trip, residual_trip);
                    if (trip == 1) break;                   // (the candidate of the adopted step size is in column 0 now)
                    J = J_trip; residual = residual_trip;
                    const float delta_J = -my_alpha * (r.dV1 + my_alpha * r.dV2);
                    const float dcost = r.J - J;
                    const float z = (delta_J > 0.0f) ? dcost / delta_J : sgnf(dcost);
                    const bool ok = gl < cfg.n_alphas && z >= cfg.c1;
                    const unsigned mask = (unsigned)((__ballot(ok) >> (grp * G)) & 0xFFFFu);
                    accept = mask != 0;
                    chosen = accept ? (__ffs(mask) - 1) : cfg.n_alphas - 1;     // first accepted, else the last tried
                    res_chosen = __shfl(residual, grp * G + chosen, 64);
                    small_step = res_chosen < cfg.atol;                          // :253-257
                    column = chosen;
                    if (!((small_step || accept) && chosen >= stored)) break;    // (group-uniform) nothing to adopt, or its candidate was kept
                    alpha_roll = cfg.alphas[chosen];
                    sink_roll = ScratchSink<N, M>{scratch, T, leader};
                    column = 0;
                }
                TFMPC_PROBE(5);
                if (a.trace.rows) {
                    const float J_chosen = __shfl(J, grp * G + chosen, 64);
                    if (leader) trace_write(a.trace, b, iteration + attempts, iteration, mu, delta, r.J, r.g_norm, chosen,
                                            cfg.alphas[chosen], J_chosen, accept ? 1 : 0, res_chosen, level);
                }
                if (small_step || accept) {
                    // adopt the chosen step size's candidate: its scratch column becomes the nominal trajectory
                    TFMPC_PROBE_START();
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const float *col = scratch + column;
                    const int rx = T * N, ru = T * M;
                    for (int rr = gl; rr < rx; rr += G) st.at(N + rr) = col[rr * ScratchSink<N, M>::kStride];
                    for (int rr = gl; rr < ru; rr += G) st.at(st.uoff() + rr) = col[(rx + rr) * ScratchSink<N, M>::kStride];
                    for (int rr = gl; rr <= T; rr += G) chat[rr] = col[(rx + ru + rr) * ScratchSink<N, M>::kStride];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    TFMPC_PROBE(6);
                }
                if (small_step) {
                    finished = true;
                } else if (accept) {                                             // :259-266: the next iteration of :227
                    delta = fminf(1.0f / cfg.delta_0, delta / cfg.delta_0);
                    mu = (mu * delta > cfg.mu_min) ? mu * delta : 0.0f;
                    need_pre = true;
                    if (++iteration >= cfg.max_iterations) { iteration = cfg.max_iterations - 1; finished = true; }   // python's loop variable after exhaustion
                } else {
                    delta = fmaxf(cfg.delta_0, delta * cfg.delta_0);             // :267-270: once more from :238
                    mu = fmaxf(cfg.mu_min, mu * delta);
                    if (++attempts >= cfg.max_attempts || !(mu < 1e30f)) { give_up = true; finished = true; }
                }
            }
            if (finished) {
                if (give_up) status |= TFMPC_ST_MAX_ATTEMPTS;
                for (int idx = gl; idx < (T + 1) * N; idx += G) xout[idx] = st.at(idx);
                for (int idx = gl; idx < T * M; idx += G) uout[idx] = st.at(st.uoff() + idx);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // chat[T] may have been written by another lane
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (leader) {
                    const float cT = chat[T];
                    if (!(cT == cT)) status |= TFMPC_ST_NAN;
                    a.iterations[b] = iteration;
                    a.status[b] = status;
#ifdef TFMPC_PHASE_PROBE
                    for (int i = 0; i < 8; ++i) chat[i] = (float)pr.acc[i];
#endif
                }
                have = false;
            }
        }
    }
}

}  // namespace tfmpc
