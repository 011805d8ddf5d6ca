// ilqr_wave_kernels.h -- the wave-per-instance iLQR kernels as templates over the env kind: model providers for the backward pass, rollout
// (iLQR.start), derivatives, backward, forward and the fused whole-solve kernel.  Device code only, so that it can be instantiated both by
// the library (ilqr_kernels.hip: the five built-in envs) and by a translation unit compiled at run time around a USER's device functions
// (user_env.h / tfmpc.envs.deviceenv: Env<TFMPC_ENV_USER>).  One wavefront per problem instance; see ilqr_core.h for the per-step arithmetic.
#pragma once

#include <hip/hip_runtime.h>

#include "ilqr_lq_mfma.h"      // kIlqrRetryBit
#include "ilqr_core.h"
#include "ilqr_trace.h"

namespace tfmpc {

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
    float *spec_extra;                      // user envs on the costate path: three more candidate trajectories per instance (user_env.h), or null
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
    int last_index = 0;                      // (user envs on the costate path: the step size the previous line search accepted)
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
            bool searched = false;
            int adopt_slot = 1;                          // (user envs on the costate path: where the candidate to adopt is, see speculative_search)
            float *adopt_extra = nullptr;
            if constexpr (kAdjoint && KIND == TFMPC_ENV_USER) {
                // a user env on the costate path: every step size at once, one per lane (user_env.h: speculative_search); the candidate of the
                // guessed step size is stored on the way, any other chosen one is rolled out again below -- the same arithmetic per step size
                if (!cfg.storage_bf16 && cfg.n_alphas <= kWave) {
                    int chosen = 0;
                    float *extra = a.spec_extra ? a.spec_extra + (size_t)b * (Env<KIND>::kSlots - 1) * Env<KIND>::candidate_floats(T) : nullptr;
                    Env<KIND>::speculative_search(e, cfg, T, xhat, uhat, kg, r.J, r.dV1, last_index < cfg.n_alphas ? last_index : 0, xc, uc, cc, extra, chosen, accept, J,
                                                  residual, adopt_slot);
                    wsync();
                    if (adopt_slot < 0) {               // not among the stored candidates: once more
                        forward_pass<KIND, false>(s, e, T, cfg.alphas[chosen], xhat, uhat, Kg, kg, xc, uc, cc, J, residual);
                        wsync();
                        adopt_slot = Env<KIND>::kSlotOfGuess;
                    }
                    adopt_extra = extra;
                    ai_last = chosen;
                    if (accept) last_index = chosen;
                    searched = true;
                }
            }
            for (int ai = 0; !searched && ai < cfg.n_alphas; ++ai) {
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
                const float *xf = xc, *uf = uc, *cf = cc;
                if constexpr (kAdjoint && KIND == TFMPC_ENV_USER) {
                    float *xs_, *us_, *cs_;
                    Env<KIND>::slot_pointers(adopt_slot, T, xc, uc, cc, adopt_extra, xs_, us_, cs_);
                    xf = xs_; uf = us_; cf = cs_;
                }
                for (int idx = lane; idx < (T + 1) * n; idx += kWave) xhat[idx] = xf[idx];
                for (int idx = lane; idx < T * m; idx += kWave) uhat[idx] = uf[idx];
                for (int idx = lane; idx <= T; idx += kWave) chat[idx] = cf[idx];
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

}  // namespace tfmpc
