// ilqr_kernels.hip -- gfx950 kernels and extern "C" entry points of the iLQR path
// (include/tfmpc_hip.h): rollout (iLQR.start), derivatives, backward, forward, the
// fused whole-solve kernel and the stand-alone box-QP.  One wavefront per problem
// instance throughout; see ilqr_core.h for the per-step arithmetic.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "ilqr_adjoint.h"
#include "ilqr_core.h"
#include "ilqr_lq_mfma.h"
#include "ilqr_trace.h"
#include "lqr_kernels.h"
#include "options.h"

namespace tfmpc {

// lane-per-instance fused solve for tiny envs (ilqr_lane.hip)
bool ilqr_lane_supported(const TfmpcEnv &env);
int ilqr_lane_solve_launch(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, int B, int T, const float *x0,
                           const float *u_init, float *states, float *actions, float *costs, int32_t *iterations,
                           int32_t *status, float *wsK, float *wsk, float *wsx, float *wsu, float *wsc, void *extra,
                           const TraceArgs &trace, hipStream_t stream);
size_t ilqr_lane_extra_workspace_bytes(int B, int n, int m, int T);
bool ilqr_lane_group_fits(int T);
int boxqp_lane2_launch(int B, const float *H, const float *q, const float *low, const float *high, const float *x0,
                       float *x, float *free_mask, int32_t *status, hipStream_t stream);   // m = 2: one QP per lane       // the 16-lanes-per-instance kernel (the one that records a decision trace) serves this horizon

// ---- model providers for backward_pass ------------------------------------------
template <int KIND>
struct EnvProvider {            // linearise on the fly from the nominal trajectory
    IlqrSmem &s;
    const EnvLds &e;
    const float *xhat, *uhat;   // [T+1][n], [T][m] (global)
    int T;
    __device__ float load(int t)
    {
        for (int i = lane_id(); i < s.n; i += kWave) s.xh[i] = xhat[(size_t)t * s.n + i];
        for (int a = lane_id(); a < s.m; a += kWave) s.uh[a] = uhat[(size_t)t * s.m + a];
        wsync();
        return Env<KIND>::linearize(e, s.xh, s.uh, s.fx, s.fu, s.lx, s.lu, s.lxx, s.luu, s.lux);
    }
    __device__ float load_final()
    {
        for (int i = lane_id(); i < s.n; i += kWave) s.xh[i] = xhat[(size_t)T * s.n + i];
        wsync();
        return Env<KIND>::final_quad(e, s.xh, s.Vx, s.Vxx);
    }
};

struct MaterialisedProvider {   // models already in HBM (reference API: iLQR.backward arguments)
    IlqrSmem &s;
    const float *uhat, *f_x, *f_u, *l, *l_x, *l_u, *l_xx, *l_uu, *l_xu, *fl, *fl_x, *fl_xx;
    __device__ float load(int t)
    {
        const int n = s.n, m = s.m, lane = lane_id();
        load_matrix(s.fx, s.ldn, f_x + (size_t)t * n * n, n, n);
        load_matrix(s.fu, s.ldm, f_u + (size_t)t * n * m, n, m);
        load_matrix(s.lxx, s.ldn, l_xx + (size_t)t * n * n, n, n);
        load_matrix(s.luu, s.ldm, l_uu + (size_t)t * m * m, m, m);
        for (int idx = lane; idx < n * m; idx += kWave) {          // Q_ux uses l_xu^T (ilqr.py:131)
            const int i = idx / m, a = idx - i * m;
            s.lux[a * s.ldn + i] = l_xu[(size_t)t * n * m + idx];
        }
        for (int i = lane; i < n; i += kWave) s.lx[i] = l_x[(size_t)t * n + i];
        for (int a = lane; a < m; a += kWave) { s.lu[a] = l_u[(size_t)t * m + a]; s.uh[a] = uhat[(size_t)t * m + a]; }
        return l[t];
    }
    __device__ float load_final()
    {
        load_matrix(s.Vxx, s.ldn, fl_xx, s.n, s.n);
        for (int i = lane_id(); i < s.n; i += kWave) s.Vx[i] = fl_x[i];
        return fl[0];
    }
};

// ---- kernels ---------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(kWave) void ilqr_rollout_kernel(TfmpcEnv env, int T, const float *x0, const float *actions,
                                                             float *states, float *costs)
{
    extern __shared__ float smem[];
    const int b = blockIdx.x, n = env.n, m = env.m;
    IlqrSmem s;
    float *p = ilqr_carve(s, smem, n, m);
    EnvLds e;
    env_load(e, env, b, p);
    wsync();
    rollout_pass<KIND>(s, e, T, x0 + (size_t)b * n, actions + (size_t)b * T * m, states + (size_t)b * (T + 1) * n,
                       costs + (size_t)b * (T + 1), nullptr);
}

struct DerivOut { float *f, *f_x, *f_u, *l, *l_x, *l_u, *l_xx, *l_uu, *l_ux, *l_xu, *fl, *fl_x, *fl_xx; };

template <int KIND>
__global__ __launch_bounds__(kWave) void ilqr_derivatives_kernel(TfmpcEnv env, int T, const float *states,
                                                                 const float *actions, DerivOut o)
{
    extern __shared__ float smem[];
    const int b = blockIdx.x, n = env.n, m = env.m, lane = lane_id();
    IlqrSmem s;
    float *p = ilqr_carve(s, smem, n, m);
    EnvLds e;
    env_load(e, env, b, p);
    wsync();
    EnvProvider<KIND> prov{s, e, states + (size_t)b * (T + 1) * n, actions + (size_t)b * T * m, T};
    const int ldn = s.ldn, ldm = s.ldm;
    for (int t = 0; t < T; ++t) {
        const float l = prov.load(t);
        Env<KIND>::transition(e, s.xh, s.uh, s.xn);
        wsync();
        const size_t bt = (size_t)b * T + t;
        if (o.f) for (int i = lane; i < n; i += kWave) o.f[bt * n + i] = s.xn[i];
        if (o.f_x) store_matrix(o.f_x + bt * n * n, s.fx, ldn, n, n);
        if (o.f_u) store_matrix(o.f_u + bt * n * m, s.fu, ldm, n, m);
        if (o.l && lane == 0) o.l[bt] = l;
        if (o.l_x) for (int i = lane; i < n; i += kWave) o.l_x[bt * n + i] = s.lx[i];
        if (o.l_u) for (int a = lane; a < m; a += kWave) o.l_u[bt * m + a] = s.lu[a];
        if (o.l_xx) store_matrix(o.l_xx + bt * n * n, s.lxx, ldn, n, n);
        if (o.l_uu) store_matrix(o.l_uu + bt * m * m, s.luu, ldm, m, m);
        if (o.l_ux) store_matrix(o.l_ux + bt * m * n, s.lux, ldn, m, n);
        if (o.l_xu) for (int idx = lane; idx < n * m; idx += kWave) {
            const int i = idx / m, a = idx - i * m;
            o.l_xu[bt * n * m + idx] = s.lux[a * ldn + i];
        }
        wsync();
    }
    const float fl = prov.load_final();
    wsync();
    if (o.fl && lane == 0) o.fl[b] = fl;
    if (o.fl_x) for (int i = lane; i < n; i += kWave) o.fl_x[(size_t)b * n + i] = s.Vx[i];
    if (o.fl_xx) store_matrix(o.fl_xx + (size_t)b * n * n, s.Vxx, ldn, n, n);
}

// Shared-env HVAC / Reservoir batches run 16 (n <= 16: up to 64) instances per wave on the matrix cores
// (ilqr_adjoint_mfma.hip).  For n <= 16 that kernel is the faster one at EVERY batch size, one instance included
// (n = 16, 12, 6 / 4 at B = 1 ... 256: 3.0-5.4 ms against 3.4-5.8 for the register-resident kernels).  For n > 16 one of its waves
// takes ~1.3x as long as a one-instance wave, so it pays once the register-resident kernels need a second round of
// waves (4 per SIMD x 1024 SIMDs): B = 4096 6.9 / 11.3 ms against 8.2 / 12.6, B = 6144 12.3 / 19.0 against 8.6 / 12.8
// (HVAC / Reservoir, T = 100, 12 iterations; tools/costate_mfma_check.py).  End of round 2 (wave-major buffers, bf16 operand
// split, two step sizes per pass; tools/costate_dispatch_sweep.py, n = 32): Reservoir 7.0-7.1 ms at B = 256 ... 4096
// against 8.4-11.0 for the register-resident kernel -- the 16-per-wave kernel at EVERY batch size; HVAC 6.7-6.8 ms
// against 4.5-6.75 -- the threshold stayed.  Round 3 (groups of up to eight waves, ilqr_adjoint_mfma_launch): HVAC n = 32
// 3.28-3.39 ms at B = 16 ... 4096 against 4.48-6.85 -- the 16-per-wave kernel at every batch size on both envs.
constexpr int kCostateMfmaMinBatchLarge = 1, kCostateMfmaMinBatchSmall = 1;
constexpr int kBlockedFrom = 12;         // state dimension from which the register-blocked products pay

struct BackwardArgs {
    int n, m, T, bounded;
    const float *actions, *f_x, *f_u, *l, *l_x, *l_u, *l_xx, *l_uu, *l_xu, *fl, *fl_x, *fl_xx, *low, *high, *mu;
    long mu_stride;
    float *K, *k, *J, *dV1, *dV2;
    int32_t *status;
};

template <bool BLK>
__global__ __launch_bounds__(kWave) void ilqr_backward_kernel(BackwardArgs a)
{
    extern __shared__ float smem[];
    const int b = blockIdx.x, n = a.n, m = a.m, T = a.T, lane = lane_id();
    IlqrSmem s;
    float *p = ilqr_carve(s, smem, n, m);
    float *low = p, *high = p + m;
    for (int j = lane; j < m; j += kWave) { low[j] = a.low[j]; high[j] = a.high[j]; }
    wsync();
    const size_t bT = (size_t)b * T;
    MaterialisedProvider prov{s, a.actions + bT * m, a.f_x + bT * n * n, a.f_u + bT * n * m, a.l + bT, a.l_x + bT * n,
                              a.l_u + bT * m, a.l_xx + bT * n * n, a.l_uu + bT * m * m, a.l_xu + bT * n * m,
                              a.fl + b, a.fl_x + (size_t)b * n, a.fl_xx + (size_t)b * n * n};
    const float mu = a.mu[(size_t)b * a.mu_stride];
    BackwardResult r = backward_pass<BLK>(s, prov, T, mu, a.bounded != 0, low, high, a.K + bT * m * n, a.k + bT * m);
    if (lane == 0) {
        a.J[b] = r.J; a.dV1[b] = r.dV1; a.dV2[b] = r.dV2;
        if (a.status) a.status[b] = r.flags | (r.failed ? TFMPC_ST_NOT_PD : 0);
    }
}

template <int KIND>
__global__ __launch_bounds__(kWave) void ilqr_forward_kernel(TfmpcEnv env, int T, const float *x, const float *u,
                                                             const float *K, const float *k, const float *alpha,
                                                             long alpha_stride, float *states, float *actions,
                                                             float *costs, float *J, float *residual)
{
    extern __shared__ float smem[];
    const int b = blockIdx.x, n = env.n, m = env.m;
    IlqrSmem s;
    float *p = ilqr_carve(s, smem, n, m);
    EnvLds e;
    env_load(e, env, b, p);
    wsync();
    float Jv, rv;
    forward_pass<KIND>(s, e, T, alpha[(size_t)b * alpha_stride], x + (size_t)b * (T + 1) * n, u + (size_t)b * T * m,
                       K + (size_t)b * T * m * n, k + (size_t)b * T * m, states + (size_t)b * (T + 1) * n,
                       actions + (size_t)b * T * m, costs + (size_t)b * (T + 1), Jv, rv);
    if (lane_id() == 0) { J[b] = Jv; residual[b] = rv; }
}

struct SolveArgs {
    int B, T;
    const float *x0, *u_init;
    float *states, *actions, *costs;
    int32_t *iterations, *status;
    float *wsK, *wsk, *wsx, *wsu, *wsc;     // per-instance scratch: gains and the candidate trajectory
    int only_flagged;                       // second-chance launch: solve only instances with kIlqrRetryBit set
    TraceArgs trace;                        // optional decision trace (tfmpc_ilqr_solve_trace_f32)
};

// iLQR.solve (ilqr.py:214-283): the whole iteration loop of one instance in one wave.
template <int KIND, bool BLK = false>
__global__ __launch_bounds__(kWave) void ilqr_solve_kernel(TfmpcEnv env, TfmpcIlqrConfig cfg, SolveArgs a)
{
    extern __shared__ float smem[];
    const int b = blockIdx.x, n = env.n, m = env.m, T = a.T, lane = lane_id();
    if (a.only_flagged && !(a.status[b] & kIlqrRetryBit)) return;      // wave-uniform
    constexpr bool kAdjoint = Env<KIND>::kPiecewiseLinearCost;   // HVAC / Reservoir: V_xx == 0 always
    IlqrSmem s;
    float *p = kAdjoint ? ilqr_carve_adjoint(s, smem, n, m) : ilqr_carve(s, smem, n, m);
    s.bf16 = cfg.storage_bf16;
    EnvLds e;
    env_load(e, env, b, p);
    wsync();

    float *xhat = a.states + (size_t)b * (T + 1) * n, *uhat = a.actions + (size_t)b * T * m,
          *chat = a.costs + (size_t)b * (T + 1);
    float *Kg = a.wsK + (size_t)b * T * m * n, *kg = a.wsk + (size_t)b * T * m;
    float *xc = a.wsx + (size_t)b * (T + 1) * n, *uc = a.wsu + (size_t)b * T * m, *cc = a.wsc + (size_t)b * (T + 1);

    // start (ilqr.py:218): nominal trajectory from the injected actions
    rollout_pass<KIND>(s, e, T, a.x0 + (size_t)b * n, a.u_init + (size_t)b * T * m, xhat, chat, uhat);
    wsync();

    float mu = 0.0f, delta = 1.0f;                                        // :215-216
    int status = 0, attempts = 0, iteration = 0;
    bool converged = false, give_up = false;
    const bool bounded = env.bounded != 0;
    EnvProvider<KIND> prov{s, e, xhat, uhat, T};

    for (iteration = 0; iteration < cfg.max_iterations; ++iteration) {     // :227
        for (;;) {                                                         // :238
            // _backward (:285-315): retry with a LOCAL regularisation bump on Cholesky failure
            float mu_l = mu, delta_l = delta;
            BackwardResult r;
            int level = 0;                                                 // local bumps before the sweep factorised (trace column)
            for (int retry = 0;; ++retry) {
                if constexpr (kAdjoint) r = backward_pass_adjoint<KIND>(s, e, T, xhat, uhat, kg);
                else r = backward_pass<BLK>(s, prov, T, mu_l, bounded, e.low, e.high, Kg, kg);
                status |= r.flags;
                if (!r.failed) break;
                status |= TFMPC_ST_NOT_PD;
                delta_l = fmaxf(cfg.delta_0, delta_l * cfg.delta_0);       // :308-309
                mu_l = fmaxf(cfg.mu_min, mu_l * delta_l);
                if (retry >= 40) { give_up = true; break; }
                ++level;
                wsync();
            }
            if (give_up) break;
            if (r.g_norm < cfg.atol) {                                     // :243-248
                if (lane == 0) trace_write(a.trace, b, iteration + attempts, iteration, mu, delta, r.J, r.g_norm, -1, 0.0f, 0.0f, -1, -1.0f, level);
                converged = true;
                break;
            }
            wsync();
            // _forward (:317-355): backtracking line search over the step sizes
            bool accept = false;
            float residual = 0.0f, J = 0.0f;
            int ai_last = -1;
            for (int ai = 0; ai < cfg.n_alphas; ++ai) {
                const float alpha = cfg.alphas[ai];
                ai_last = ai;
                forward_pass<KIND, !kAdjoint>(s, e, T, alpha, xhat, uhat, Kg, kg, xc, uc, cc, J, residual);
                const float delta_J = -alpha * (r.dV1 + alpha * r.dV2);    // :339
                const float dcost = r.J - J;
                const float z = (delta_J > 0.0f) ? dcost / delta_J : signf(dcost);   // :342-346
                wsync();
                if (z >= cfg.c1) { accept = true; break; }                 // :351-353
            }
            const bool small_step = residual < cfg.atol;                  // :253-257 (taken even if rejected)
            if (lane == 0)
                trace_write(a.trace, b, iteration + attempts, iteration, mu, delta, r.J, r.g_norm, ai_last,
                            ai_last >= 0 ? cfg.alphas[ai_last] : 0.0f, J, accept ? 1 : 0, residual, level);
            if (small_step || accept) {
                for (int idx = lane; idx < (T + 1) * n; idx += kWave) xhat[idx] = xc[idx];
                for (int idx = lane; idx < T * m; idx += kWave) uhat[idx] = uc[idx];
                for (int idx = lane; idx <= T; idx += kWave) chat[idx] = cc[idx];
                wsync();
            }
            if (small_step) { converged = true; break; }
            if (accept) {                                                  // :259-266
                delta = fminf(1.0f / cfg.delta_0, delta / cfg.delta_0);
                mu = (mu * delta > cfg.mu_min) ? mu * delta : 0.0f;
                break;
            }
            delta = fmaxf(cfg.delta_0, delta * cfg.delta_0);               // :267-270
            mu = fmaxf(cfg.mu_min, mu * delta);
            if (++attempts >= cfg.max_attempts || !(mu < 1e30f)) { give_up = true; break; }
        }
        if (converged || give_up) break;                                   // :276-277
    }
    if (give_up) status |= TFMPC_ST_MAX_ATTEMPTS;
    if (iteration >= cfg.max_iterations) iteration = cfg.max_iterations - 1;   // python's loop variable after exhaustion
    if (lane == 0) {
        const float c0 = chat[T];
        if (!(c0 == c0)) status |= TFMPC_ST_NAN;
        a.iterations[b] = iteration;
        a.status[b] = status;
    }
}

__global__ __launch_bounds__(kWave) void boxqp_kernel(int m, const float *H, const float *q, const float *low,
                                                      const float *high, const float *x0, float *x, float *free_mask,
                                                      int32_t *status)
{
    extern __shared__ float smem[];
    const int b = blockIdx.x, lane = lane_id();
    IlqrSmem s;
    float *p = ilqr_carve(s, smem, 1, m);
    const int ldm = s.ldm;
    float *Hm = p, *qv = p + m * ldm;
    load_matrix(Hm, ldm, H + (size_t)b * m * m, m, m);
    for (int i = lane; i < m; i += kWave) {
        qv[i] = q[(size_t)b * m + i];
        s.qlo[i] = low[(size_t)b * m + i];
        s.qhi[i] = high[(size_t)b * m + i];
        s.qx[i] = x0[(size_t)b * m + i];
    }
    wsync();
    const int rc = boxqp(s, Hm, ldm, qv);
    wsync();
    for (int i = lane; i < m; i += kWave) {
        x[(size_t)b * m + i] = s.qx[i];
        if (free_mask) free_mask[(size_t)b * m + i] = s.qfree[i];
    }
    if (status && lane == 0) status[b] = rc;
}

// ---- host side ---------------------------------------------------------------------
static size_t ilqr_smem_bytes(int kind, int n, int m, int zones)
{
    return (ilqr_smem_floats(n, m) + env_lds_floats(kind, n, m, zones) + 2 * (size_t)m) * sizeof(float);
}

// the fused solve of HVAC / Reservoir uses the adjoint backward pass and a reduced LDS slice
static size_t ilqr_solve_smem_bytes(int kind, int n, int m, int zones)
{
    const bool adjoint = kind == TFMPC_ENV_HVAC || kind == TFMPC_ENV_RESERVOIR;
    const size_t core = adjoint ? ilqr_adjoint_smem_floats(n, m) : ilqr_smem_floats(n, m);
    return (core + env_lds_floats(kind, n, m, zones) + 2 * (size_t)m) * sizeof(float);
}

static int check_env(const TfmpcEnv *env)
{
    if (!env || env->n <= 0 || env->m <= 0 || !env->low || !env->high) return TFMPC_ERR_ARG;
    if (env->kind < TFMPC_ENV_LQ || env->kind > TFMPC_ENV_RESERVOIR) return TFMPC_ERR_ARG;
    if (env->kind != TFMPC_ENV_LQ && env->n != env->m) return TFMPC_ERR_ARG;   // reference envs: action_size == state_size
    if (env->kind == TFMPC_ENV_NAVIGATION && (env->n > 8 || env->n_zones < 0)) return TFMPC_ERR_ARG;
    for (int i = 0; i < TFMPC_ENV_MAX_PARAMS; ++i)
        if (env_param_len(env->kind, i, env->n, env->m, env->n_zones) > 0 && !env->p[i]) return TFMPC_ERR_ARG;
    if (ilqr_smem_bytes(env->kind, env->n, env->m, env->n_zones) > kMaxLdsBytes) return TFMPC_ERR_UNSUPPORTED;
    return TFMPC_OK;
}

template <class Kern>
static int prep(Kern kern, size_t smem)
{
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return TFMPC_ERR_LAUNCH;
    return TFMPC_OK;
}

static int launched() { return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH; }

#define TFMPC_DISPATCH_KIND(kind, CALL)                                           \
    switch (kind) {                                                               \
    case TFMPC_ENV_LQ: { constexpr int KIND = TFMPC_ENV_LQ; CALL; } break;         \
    case TFMPC_ENV_NAVLQR: { constexpr int KIND = TFMPC_ENV_NAVLQR; CALL; } break; \
    case TFMPC_ENV_NAVIGATION: { constexpr int KIND = TFMPC_ENV_NAVIGATION; CALL; } break; \
    case TFMPC_ENV_HVAC: { constexpr int KIND = TFMPC_ENV_HVAC; CALL; } break;     \
    case TFMPC_ENV_RESERVOIR: { constexpr int KIND = TFMPC_ENV_RESERVOIR; CALL; } break; \
    default: return TFMPC_ERR_ARG;                                                \
    }

}  // namespace tfmpc

using namespace tfmpc;

extern "C" {

int tfmpc_ilqr_rollout_f32(const TfmpcEnv *env, int B, int T, const float *x0, const float *actions, float *states,
                           float *costs, void *stream)
{
    int rc = check_env(env);
    if (rc != TFMPC_OK) return rc;
    if (B < 0 || T < 0) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!x0 || !states || !costs || (T > 0 && !actions)) return TFMPC_ERR_ARG;
    const size_t smem = ilqr_smem_bytes(env->kind, env->n, env->m, env->n_zones);
    hipStream_t st = static_cast<hipStream_t>(stream);
    TFMPC_DISPATCH_KIND(env->kind, {
        auto kern = ilqr_rollout_kernel<KIND>;
        if ((rc = prep(kern, smem)) != TFMPC_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWave), smem, st, *env, T, x0, actions, states, costs);
    });
    return launched();
}

int tfmpc_ilqr_derivatives_f32(const TfmpcEnv *env, int B, int T, const float *states, const float *actions, float *f,
                               float *f_x, float *f_u, float *l, float *l_x, float *l_u, float *l_xx, float *l_uu,
                               float *l_ux, float *l_xu, float *fl, float *fl_x, float *fl_xx, void *stream)
{
    int rc = check_env(env);
    if (rc != TFMPC_OK) return rc;
    if (B < 0 || T < 0) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!states || (T > 0 && !actions)) return TFMPC_ERR_ARG;
    const size_t smem = ilqr_smem_bytes(env->kind, env->n, env->m, env->n_zones);
    hipStream_t st = static_cast<hipStream_t>(stream);
    DerivOut o{f, f_x, f_u, l, l_x, l_u, l_xx, l_uu, l_ux, l_xu, fl, fl_x, fl_xx};
    TFMPC_DISPATCH_KIND(env->kind, {
        auto kern = ilqr_derivatives_kernel<KIND>;
        if ((rc = prep(kern, smem)) != TFMPC_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWave), smem, st, *env, T, states, actions, o);
    });
    return launched();
}

int tfmpc_ilqr_backward_f32(int B, int n, int m, int T, const float *actions, const float *f_x, const float *f_u,
                            const float *l, const float *l_x, const float *l_u, const float *l_xx, const float *l_uu,
                            const float *l_xu, const float *fl, const float *fl_x, const float *fl_xx,
                            const float *low, const float *high, int bounded, const float *mu, long mu_stride,
                            float *K, float *k, float *J, float *dV1, float *dV2, int32_t *status, void *stream)
{
    if (B < 0 || n <= 0 || m <= 0 || T < 0) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!fl || !fl_x || !fl_xx || !low || !high || !mu || !J || !dV1 || !dV2) return TFMPC_ERR_ARG;
    if (T > 0 && (!actions || !f_x || !f_u || !l || !l_x || !l_u || !l_xx || !l_uu || !l_xu || !K || !k)) return TFMPC_ERR_ARG;
    const size_t smem = (ilqr_smem_floats(n, m) + 2 * (size_t)m) * sizeof(float);
    if (smem > kMaxLdsBytes) return TFMPC_ERR_UNSUPPORTED;
    const bool blk = n >= kBlockedFrom;
    int rc = blk ? prep(ilqr_backward_kernel<true>, smem) : prep(ilqr_backward_kernel<false>, smem);
    if (rc != TFMPC_OK) return rc;
    BackwardArgs a{n, m, T, bounded, actions, f_x, f_u, l, l_x, l_u, l_xx, l_uu, l_xu, fl, fl_x, fl_xx, low, high, mu,
                   mu_stride, K, k, J, dV1, dV2, status};
    if (blk) hipLaunchKernelGGL(ilqr_backward_kernel<true>, dim3(B), dim3(kWave), smem, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(ilqr_backward_kernel<false>, dim3(B), dim3(kWave), smem, static_cast<hipStream_t>(stream), a);
    return launched();
}

int tfmpc_ilqr_forward_f32(const TfmpcEnv *env, int B, int T, const float *x, const float *u, const float *K,
                           const float *k, const float *alpha, long alpha_stride, float *states, float *actions,
                           float *costs, float *J, float *residual, void *stream)
{
    int rc = check_env(env);
    if (rc != TFMPC_OK) return rc;
    if (B < 0 || T < 0) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!x || !alpha || !states || !costs || !J || !residual) return TFMPC_ERR_ARG;
    if (T > 0 && (!u || !K || !k || !actions)) return TFMPC_ERR_ARG;
    const size_t smem = ilqr_smem_bytes(env->kind, env->n, env->m, env->n_zones);
    hipStream_t st = static_cast<hipStream_t>(stream);
    TFMPC_DISPATCH_KIND(env->kind, {
        auto kern = ilqr_forward_kernel<KIND>;
        if ((rc = prep(kern, smem)) != TFMPC_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWave), smem, st, *env, T, x, u, K, k, alpha, alpha_stride, states,
                           actions, costs, J, residual);
    });
    return launched();
}

// The ONE description of the solve workspace (tfmpc_ilqr_workspace_bytes sizes it, tfmpc_ilqr_solve_trace_f32 carves it):
// five per-instance slabs -- gains K[T][m][n], k[T][m] and a candidate trajectory x[T+1][n], u[T][m], c[T+1] -- then, each
// from a 256-byte boundary, the line-search scratch of the 2-D lane-group kernel and the wave-major trajectory buffers of
// the 16-instances-per-wave HVAC / Reservoir kernel.  Offsets in bytes from a 256-byte aligned base.
struct IlqrWsLayout {
    size_t K, k, x, u, c, lane, wave, total;
    IlqrWsLayout(int B, int n, int m, int T)
    {
        const size_t f = sizeof(float);
        K = 0;
        k = K + (size_t)B * T * m * n * f;
        x = k + (size_t)B * T * m * f;
        u = x + (size_t)B * (T + 1) * n * f;
        c = u + (size_t)B * T * m * f;
        lane = round256(c + (size_t)B * (T + 1) * f);
        wave = lane + round256(ilqr_lane_extra_workspace_bytes(B, n, m, T));
        total = wave + ilqr_adjoint_mfma_workspace_bytes(B, n, m, T);
    }
    static size_t round256(size_t v) { return (v + 255) & ~(size_t)255; }
};

size_t tfmpc_ilqr_workspace_bytes(int B, int n, int m, int T)
{
    if (B <= 0 || n <= 0 || m <= 0 || T < 0) return 0;
    return IlqrWsLayout(B, n, m, T).total;
}

// which kernel family the last tfmpc_ilqr_solve[_trace]_f32 of this thread went to (a traced solve of HVAC / Reservoir, or of a
// tiny 2-D env forced onto the per-lane kernel, takes a kernel that records a trace: the caller can see which one solved it)
static thread_local const char *g_last_ilqr_kernel = "";

const char *tfmpc_ilqr_last_kernel_name(void) { return g_last_ilqr_kernel; }

int tfmpc_ilqr_solve_f32(const TfmpcEnv *env, const TfmpcIlqrConfig *cfg, int B, int T, const float *x0,
                         const float *u_init, float *states, float *actions, float *costs, int32_t *iterations,
                         int32_t *status, void *workspace, size_t workspace_bytes, void *stream)
{
    return tfmpc_ilqr_solve_trace_f32(env, cfg, B, T, x0, u_init, states, actions, costs, iterations, status, nullptr, 0,
                                      nullptr, workspace, workspace_bytes, stream);
}

int tfmpc_ilqr_solve_trace_f32(const TfmpcEnv *env, const TfmpcIlqrConfig *cfg, int B, int T, const float *x0,
                               const float *u_init, float *states, float *actions, float *costs, int32_t *iterations,
                               int32_t *status, float *trace, int trace_rows, int32_t *trace_len, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    int rc = check_env(env);
    if (rc != TFMPC_OK) return rc;
    if (!cfg || B < 0 || T < 0 || cfg->n_alphas < 1 || cfg->n_alphas > TFMPC_MAX_ALPHAS || cfg->max_iterations < 1)
        return TFMPC_ERR_ARG;
    if (trace && (trace_rows < 1 || !trace_len)) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!x0 || !states || !costs || !iterations || !status || (T > 0 && (!u_init || !actions))) return TFMPC_ERR_ARG;
    const int n = env->n, m = env->m;
    const IlqrWsLayout lay(B, n, m, T);
    if (!workspace || workspace_bytes < lay.total || (reinterpret_cast<uintptr_t>(workspace) & 255u)) return TFMPC_ERR_WORKSPACE;
    char *const base = static_cast<char *>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const TraceArgs tr{trace, trace_len, trace ? trace_rows : 0};
    const bool traced = trace != nullptr;
    if (traced && hipMemsetAsync(trace_len, 0, (size_t)B * sizeof(int32_t), st) != hipSuccess) return TFMPC_ERR_LAUNCH;
    SolveArgs a{};
    a.B = B; a.T = T; a.x0 = x0; a.u_init = u_init;
    a.states = states; a.actions = actions; a.costs = costs; a.iterations = iterations; a.status = status;
    a.wsK = reinterpret_cast<float *>(base + lay.K);
    a.wsk = reinterpret_cast<float *>(base + lay.k);
    a.wsx = reinterpret_cast<float *>(base + lay.x);
    a.wsu = reinterpret_cast<float *>(base + lay.u);
    a.wsc = reinterpret_cast<float *>(base + lay.c);
    a.trace = tr;
    char *const after_slabs = base + lay.lane;
    char *const after_lane = base + lay.wave;
    {
        // tiny 2-D envs: 16 lanes per instance with a speculative parallel line search (ilqr_lane.hip) at EVERY batch
        // size -- also for one instance it has the shorter critical path (Navigation, T = 50, B = 1: 2.3 ms against
        // 7.2 ms for the wave kernel; tools/small_batch_lane_vs_wave.py).  TFMPC_ILQR_KERNEL=lane|lane1|wave forces.
        const bool forced_wave = option_is(kOptIlqrKernel, "wave");
        if (ilqr_lane_supported(*env) && !forced_wave && !cfg->storage_bf16 && (!traced || ilqr_lane_group_fits(T))) {
            g_last_ilqr_kernel = (option_is(kOptIlqrKernel, "lane1") && !traced) ? "lane (one lane per instance)" : "lane_group (16 lanes per instance, instance queue)";
            return ilqr_lane_solve_launch(*env, *cfg, B, T, x0, u_init, states, actions, costs, iterations, status,
                                          a.wsK, a.wsk, a.wsx, a.wsu, a.wsc,
                                          after_slabs, tr, st);
        }
    }
    {
        // LQ env on the matrix cores (ilqr_lq_mfma.hip); instances it cannot finish (mu > 0 needed)
        // come back flagged and are re-solved from scratch by the wave kernel right behind it
        // (a traced solve stays on these kernels: each records the same row per pass as the wave kernel, and the
        // second-chance launch rewrites the rows of the instances it re-solves)
        const bool forced_wave = option_is(kOptIlqrKernel, "wave");
        if (!forced_wave && !cfg->storage_bf16 && ilqr_lq_box_mfma_supported(*env, T)) {
            // control-limited LQ problems: box-QP in registers, the complete solve loop in one kernel
            IlqrLqArgs la{*env, *cfg, B, T, x0, u_init, states, actions, costs, iterations, status, a.wsK, a.wsk, a.wsu};
            la.trace = tr;
            g_last_ilqr_kernel = "lq_box_mfma (matrix cores, control-limited)";
            return ilqr_lq_box_mfma_launch(la, st);
        }
        if (!forced_wave && !cfg->storage_bf16 && ilqr_lq_mfma32_supported(*env, T)) {
            // BASELINE configs[4]'s literal dims (n <= 32, m <= 16): 2 x 2 tiles, trajectories in HBM; second chance as below
            IlqrLqArgs la{*env, *cfg, B, T, x0, u_init, states, actions, costs, iterations, status, a.wsK, a.wsk, a.wsu,
                          a.wsx, a.wsu, a.wsc};
            la.trace = tr;
            if ((rc = ilqr_lq_mfma32_launch(la, st)) != TFMPC_OK) return rc;
            a.only_flagged = 1;
            g_last_ilqr_kernel = "lq_mfma32 (matrix cores, 2 x 2 tiles) + wave kernel for flagged instances";
        }
        if (!forced_wave && !cfg->storage_bf16 && ilqr_lq_mfma_supported(*env, T)) {
            IlqrLqArgs la{*env, *cfg, B, T, x0, u_init, states, actions, costs, iterations, status, a.wsK, a.wsk, a.wsu};
            la.trace = tr;
            if ((rc = ilqr_lq_mfma_launch(la, st)) != TFMPC_OK) return rc;
            a.only_flagged = 1;
            g_last_ilqr_kernel = "lq_mfma (matrix cores) + wave kernel for flagged instances";
        }
    }
    {
        // HVAC / Reservoir at n <= 32: register-resident costate kernel (ilqr_adjoint.hip),
        // bit-identical to the generic wave kernel below (TFMPC_ILQR_KERNEL=wave selects that one)
        const bool forced_wave = option_is(kOptIlqrKernel, "wave");
        const AdjointSolveArgs aa{B, T, x0, u_init, states, actions, costs, iterations, status, a.wsk, a.wsx, a.wsu, a.wsc, after_lane, tr,
                                  option_is(kOptCostateCoupling, "dense") ? 1 : 0};
        // a batch that shares one env: 16 instances per wave with the coupling-matrix products on the matrix
        // cores (ilqr_adjoint_mfma.hip); TFMPC_ILQR_KERNEL=costate_mfma forces it, lean / lean1 the kernels above.
        // storage_bf16: that kernel keeps its trajectories in REAL 16-bit containers (the wave kernel below emulates
        // the format in fp32 containers and serves every env the 16-per-wave kernel does not)
        const bool forced_lean = option_is(kOptIlqrKernel, "lean") || option_is(kOptIlqrKernel, "lean1");
        const bool forced_mfma = option_is(kOptIlqrKernel, "costate_mfma");
        if (!forced_wave && !forced_lean && ilqr_adjoint_mfma_supported(*env, *cfg) &&
            (forced_mfma || cfg->storage_bf16 || traced ||
             B >= ((n > 16 && env->kind != TFMPC_ENV_RESERVOIR) ? kCostateMfmaMinBatchLarge : kCostateMfmaMinBatchSmall)))
        {
            g_last_ilqr_kernel = "costate_mfma (16 instances per wave)";
            return ilqr_adjoint_mfma_launch(*env, *cfg, aa, st);
        }
        if (!forced_wave && !traced && ilqr_adjoint_supported(*env, *cfg)) {
            g_last_ilqr_kernel = "costate (register-resident, one instance per wave)";
            return ilqr_adjoint_launch(*env, *cfg, aa, st);
        }
    }
    const size_t smem = ilqr_solve_smem_bytes(env->kind, n, m, env->n_zones);
    if (!a.only_flagged) g_last_ilqr_kernel = "wave (one instance per wave, any env)";
    if (env->kind == TFMPC_ENV_LQ && n >= kBlockedFrom) {      // the only dense env that comes in large shapes
        auto kern = ilqr_solve_kernel<TFMPC_ENV_LQ, true>;
        if ((rc = prep(kern, smem)) != TFMPC_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWave), smem, st, *env, *cfg, a);
        return launched();
    }
    TFMPC_DISPATCH_KIND(env->kind, {
        auto kern = ilqr_solve_kernel<KIND>;
        if ((rc = prep(kern, smem)) != TFMPC_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWave), smem, st, *env, *cfg, a);
    });
    return launched();
}

int tfmpc_boxqp_f32(int B, int m, const float *H, const float *q, const float *low, const float *high, const float *x0,
                    float *x, float *free_mask, int32_t *status, void *stream)
{
    if (B < 0 || m <= 0) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!H || !q || !low || !high || !x0 || !x) return TFMPC_ERR_ARG;
    if (m == 2 && !option_is(kOptIlqrKernel, "wave"))       // the lane kernels' own QP (closed form for two variables)
        return boxqp_lane2_launch(B, H, q, low, high, x0, x, free_mask, status, static_cast<hipStream_t>(stream));
    const size_t smem = (ilqr_smem_floats(1, m) + (size_t)m * odd_ld(m) + m) * sizeof(float);
    if (smem > kMaxLdsBytes) return TFMPC_ERR_UNSUPPORTED;
    int rc = prep(boxqp_kernel, smem);
    if (rc != TFMPC_OK) return rc;
    hipLaunchKernelGGL(boxqp_kernel, dim3(B), dim3(kWave), smem, static_cast<hipStream_t>(stream), m, H, q, low, high,
                       x0, x, free_mask, status);
    return launched();
}

}  // extern "C"
