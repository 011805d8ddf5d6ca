// user_env_group.h -- iLQR.solve (ilqr.py:214-355) for a USER env on the costate path (user_env.h: TFMPC_USER_ZERO_HESSIAN -- a piecewise-linear cost
// and bounded actions, so that the backward pass is the costate recursion, ilqr.py:137-141) with SIXTEEN LANES per instance, four
// instances per wavefront (round 6).  Included by user_env_kernels.hip.in after user_env.h / ilqr_wave_kernels.h.
//
// On that path the wave-per-instance kernel (ilqr_wave_kernels.h) keeps n + m lanes of 64 busy in its backward pass (one first-order dual evaluation of
// the user's transition and cost per direction of z = [x; u], one direction per lane) and n_alphas of 64 in its line search (every step size at once,
// one per lane): a Python-defined Reservoir / HVAC of the reference's own size (n + m = 8 / 12, 11 step sizes) runs at an eighth of the lanes, and a
// batch of 16 384 is bound by the instruction count of 16 384 waves.  Here the same two programs run in the 16-lane ROWS of a wave -- n + m <= 16,
// n_alphas <= 16 -- with the instance's state, action and costate replicated in the registers of its row:
//   * backward: lane j < n + m of a row evaluates direction j (Env<USER>::adjoint_direction, the function the wave kernel calls), Q_x comes back to every
//     lane of the row by a row-wide shuffle, the sums / maxima over the actions are the row part of wave_sum / wave_max (the same four DPP steps: in the
//     wave kernel the other three rows hold exact zeros), so dV1, g_norm, J_hat are the wave kernel's bits;
//   * line search: lane s of a row rolls out step size s, the lanes around the step size accepted last time store their candidates on the way (four
//     slots, see kUserGroupSlots), a ballot within the row picks the first step size that passes (ilqr.py:322-353); a choice outside the stored ones is
//     rolled out once more by the whole row, lane 0 storing;
//   * the solve state machine is the wave kernel's, replicated per row; rows that have finished idle until the wave's last one has.
// Arithmetic per instance is that of the wave kernel, operation for operation: outputs and decision traces are bit-identical (tests/test_fxenv_gpu.py).
// G = 32 (16 < n + m <= 32, n_alphas <= 32): the same kernel with TWO ROWS per instance, two instances per wave -- the sums / maxima take wave_sum's
// fifth step as well (rows 1, 3 += rows 0, 2) and the instance's total is read from the last lane of its second row.
#pragma once

#include "ilqr_wave_kernels.h"

namespace tfmpc {

template <int N, int M, int G_ = 16>
struct UserGroup {
    static constexpr int G = G_, D = N + M;
    static_assert(G == 16 || G == 32, "a row of sixteen lanes, or two");
    static_assert(D <= G, "one direction of z = [x; u] per lane of a group");

    // a value of the last lane of each lane's own group of 32
    static __device__ __forceinline__ unsigned of_last_lane(unsigned v)
    {
        const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 31), b = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
        return lane_id() < 32 ? a : b;
    }
    // sum over the group (the first four / five steps of wave_sum)
    static __device__ __forceinline__ float row_sum(float v)
    {
        v += dpp_move<kDppQuadXor1>(0.0f, v);
        v += dpp_move<kDppQuadXor2>(0.0f, v);
        v += dpp_move<kDppRowHalfMirror>(0.0f, v);
        v += dpp_move<kDppRowMirror>(0.0f, v);
        if constexpr (G == 32) {
            v += dpp_move<kDppRowBcast15, 0xA>(0.0f, v);           // rows 1, 3 += rows 0, 2
            v = __builtin_bit_cast(float, of_last_lane(__builtin_bit_cast(unsigned, v)));
        }
        return v;
    }
    // maximum over the group of non-negative values (the first four / five steps of wave_max)
    static __device__ __forceinline__ float row_max(float v)
    {
        unsigned u = __builtin_bit_cast(unsigned, v);
        auto step = [&](auto mv) { const unsigned o = (unsigned)mv; u = u > o ? u : o; };
        step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppQuadXor1, 0xF, 0xF, false));
        step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppQuadXor2, 0xF, 0xF, false));
        step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppRowHalfMirror, 0xF, 0xF, false));
        step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppRowMirror, 0xF, 0xF, false));
        if constexpr (G == 32) {
            step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppRowBcast15, 0xA, 0xF, false));
            u = of_last_lane(u);
        }
        return __builtin_bit_cast(float, u);
    }
};

// One rollout per lane from x0 under u_t = clip(u_hat_t + alpha k_t) (ilqr.py:174-212 with K == 0; start: the injected actions as they are) -- the
// program of Env<USER>::speculative_search / forward_pass<USER, false> / rollout_pass, operation for operation.  `keep`: this lane stores its trajectory.
template <int N, int M, bool START>
__device__ __forceinline__ void user_group_rollout(const float *p, const float *low, const float *high, int T, float alpha, const float *x0,
                                                   const float *uhat, const float *kg, bool keep, float *xs, float *us, float *cs, float &J_out,
                                                   float &rmax_out)
{
    float x[N], xn[N], u[M];
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] = x0[i]; if (keep) xs[i] = x[i]; }
    float J = 0.0f, rmax = 0.0f;
    // (the inputs of step t + 1 are requested while step t is evaluated: a register ring of depth one, refilled unconditionally with a clamped index)
    float uh_n[M], k_n[M];
#pragma unroll
    for (int a = 0; a < M; ++a) {
        uh_n[a] = T > 0 ? uhat[a] : 0.0f;
        k_n[a] = (!START && T > 0) ? kg[a] : 0.0f;
    }
    for (int t = 0; t < T; ++t) {
        const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
        for (int a = 0; a < M; ++a) {
            const float uh_c = uh_n[a], k_c = k_n[a];
            uh_n[a] = uhat[(size_t)tn * M + a];
            if constexpr (!START) k_n[a] = kg[(size_t)tn * M + a];
            if constexpr (START) {
                u[a] = uh_c;                                                                             // ilqr.py:53-82
            } else {
                const float du = alpha * k_c;                                                            // :193-194 (K == 0)
                u[a] = fminf(fmaxf(uh_c + du, low[a]), high[a]);                                         // :196-197
                rmax = fmaxf(rmax, fabsf(du));                                                            // :206
            }
            if (keep) us[(size_t)t * M + a] = u[a];
        }
        const float c = tfmpc_user::cost<float>(p, x, u);                                                // :198
        tfmpc_user::transition<float>(p, x, u, xn);                                                      // :199
        J += c;                                                                                          // :205
        if (keep) cs[t] = c;
#pragma unroll
        for (int i = 0; i < N; ++i) { x[i] = xn[i]; if (keep) xs[(size_t)(t + 1) * N + i] = xn[i]; }
    }
    const float fc = tfmpc_user::final_cost<float>(p, x);                                                // :208-210
    if (keep) cs[T] = fc;
    J_out = J + fc;
    rmax_out = rmax;
}

// Candidates of the line search: the lanes of the step sizes guess - 1 .. guess + 2 (guess = the index accepted last) store theirs, in four slots per
// instance -- slot 1 is the candidate slab every solve kernel has (wsx, wsu, wsc), slots 0, 2, 3 the extra part of this kernel's workspace.  The index
// moves between passes (res4: -1 / 0 / +1 / +2 in 8 / 23 / 37 / 17 % of the passes, hvac6 18 / 17 / 11 / 7 %; tools/probes/r6_alpha_moves.py): with
// the guess alone stored, four passes of five rolled the chosen step size out a second time -- a third of an iteration's work.
constexpr int kUserGroupSlots = 4, kUserGroupSlotOfGuess = 1;
__host__ __device__ inline size_t user_group_candidate_floats(int n, int m, int T) { return (size_t)(T + 1) * n + (size_t)T * m + (size_t)(T + 1); }
__host__ __device__ inline size_t user_group_extra_bytes(int B, int n, int m, int T)
{
    return (size_t)B * (kUserGroupSlots - 1) * user_group_candidate_floats(n, m, T) * sizeof(float);
}

template <int N, int M, int LANES = 16>
__global__ __launch_bounds__(kWave) void ilqr_user_costate_group_kernel(TfmpcEnv env, TfmpcIlqrConfig cfg, SolveArgs a, float *extra)
{
    using UG = UserGroup<N, M, LANES>;
    constexpr int G = UG::G, D = UG::D, GROUPS = kWave / G;
    extern __shared__ float smem[];
    const int lane = lane_id(), grp = lane / G, gl = lane % G, T = a.T;
    const int b_raw = (int)blockIdx.x * GROUPS + grp;
    const bool live = b_raw < a.B;                     // (a row beyond the batch repeats the last instance and stores nothing)
    const int b = live ? b_raw : a.B - 1;
    const bool leader = gl == 0;
    // the instance's parameter floats and the action bounds in LDS: [row][P] | low[M] | high[M]
    const int P = env.n_zones;
    float *pr = smem + (size_t)grp * P;
    for (int k = gl; k < P; k += G) pr[k] = env.p[0][(size_t)b * env.stride[0] + k];
    float *low = smem + (size_t)GROUPS * P, *high = low + M;
    if (lane < M) { low[lane] = env.low[lane]; high[lane] = env.high[lane]; }
    wsync();
    EnvLds e{};
    e.n = N; e.m = M; e.zones = P; e.p[0] = pr; e.low = low; e.high = high;

    float *xhat = a.states + (size_t)b * (T + 1) * N, *uhat = a.actions + (size_t)b * T * M, *chat = a.costs + (size_t)b * (T + 1);
    float *kg = a.wsk + (size_t)b * T * M;
    float *xc = a.wsx + (size_t)b * (T + 1) * N, *uc = a.wsu + (size_t)b * T * M, *cc = a.wsc + (size_t)b * (T + 1);
    const size_t candF = user_group_candidate_floats(N, M, T);
    float *const extra_b = extra + (size_t)b * (kUserGroupSlots - 1) * candF;
    auto slot_x = [&](int s_) { return s_ == kUserGroupSlotOfGuess ? xc : extra_b + (size_t)(s_ - (s_ > kUserGroupSlotOfGuess ? 1 : 0)) * candF; };
    auto slot_u = [&](int s_) { return s_ == kUserGroupSlotOfGuess ? uc : slot_x(s_) + (size_t)(T + 1) * N; };
    auto slot_c = [&](int s_) { return s_ == kUserGroupSlotOfGuess ? cc : slot_u(s_) + (size_t)T * M; };
    const int row0 = grp * G;                          // first lane of this row

    // start (ilqr.py:218): the nominal trajectory from the injected actions -- every lane of the row the same program, lane 0 stores
    {
        float J0, r0;
        user_group_rollout<N, M, true>(pr, low, high, T, 0.0f, a.x0 + (size_t)b * N, a.u_init + (size_t)b * T * M, nullptr, leader && live, xhat, uhat, chat, J0, r0);
    }
    wsync();

    float mu = 0.0f, delta = 1.0f;                                        // :215-216
    int status = 0, attempts = 0, iteration = 0, last_index = 0;
    bool converged = false, give_up = false, finished = false;

    while (__any(!finished)) {                                            // (the rows of a wave leave together)
        if (!finished) {
            // ---- backward (:94-172 on this path: the costate recursion) ----------------------------------------------------------------------
            float x[N], u[M], Vx[N];
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = xhat[(size_t)T * N + i];
            float J_hat;
            {   // V_x = l_x^f (:101): lane j < N its entry j, then to every lane of the row
                using ad::D1;
                D1 xs[N];
#pragma unroll
                for (int i = 0; i < N; ++i) xs[i] = D1(x[i], i == gl ? 1.0f : 0.0f);
                const D1 c = tfmpc_user::final_cost<D1>(pr, xs);
                J_hat = c.v;
#pragma unroll
                for (int i = 0; i < N; ++i) Vx[i] = __shfl(c.d, row0 + i, kWave);
            }
            float dV1 = 0.0f, gsum = 0.0f;
            // (the nominal point of step t - 1 is requested while step t is evaluated: a register ring of depth one, refilled unconditionally with a clamped index)
            float xn_[N], un_[M];
            if (T > 0) {
#pragma unroll
                for (int i = 0; i < N; ++i) xn_[i] = xhat[(size_t)(T - 1) * N + i];
#pragma unroll
                for (int aa = 0; aa < M; ++aa) un_[aa] = uhat[(size_t)(T - 1) * M + aa];
            }
            for (int t = T - 1; t >= 0; --t) {
#pragma unroll
                for (int i = 0; i < N; ++i) x[i] = xn_[i];
#pragma unroll
                for (int aa = 0; aa < M; ++aa) u[aa] = un_[aa];
                const int tp = t > 0 ? t - 1 : 0;
#pragma unroll
                for (int i = 0; i < N; ++i) xn_[i] = xhat[(size_t)tp * N + i];
#pragma unroll
                for (int aa = 0; aa < M; ++aa) un_[aa] = uhat[(size_t)tp * M + aa];
                float acc;
                const float l = Env<TFMPC_ENV_USER>::adjoint_direction(e, x, u, Vx, gl, acc);
                float p1 = 0.0f, gmax = 0.0f;
                if (gl >= N && gl < D) {
                    const int aa = gl - N;
                    const float kt = (acc >= 0.0f) ? (low[aa] - u[aa]) : (high[aa] - u[aa]);              // :140-141
                    if (live) kg[(size_t)t * M + aa] = kt;
                    p1 = fmaf(kt, acc, p1);
                    gmax = fmaxf(gmax, fabsf(kt) / (fabsf(u[aa]) + 1.0f));
                }
                J_hat += l;
                dV1 += UG::row_sum(p1);
                gsum += UG::row_max(gmax);
#pragma unroll
                for (int i = 0; i < N; ++i) Vx[i] = __shfl(acc, row0 + i, kWave);                          // V_x <- Q_x
            }
            const float g_norm = T > 0 ? gsum / (float)T : 0.0f;
            wsync();                                    // (the gains, written by lanes N .. D - 1, are read by every lane of the row below)
            if (g_norm < cfg.atol) {                                           // :243-248
                if (leader && live) trace_write(a.trace, b, iteration + attempts, iteration, mu, delta, J_hat, g_norm, -1, 0.0f, 0.0f, -1, -1.0f, 0);
                converged = true;
            } else {
                // ---- forward (:317-355): every step size at once, one per lane of the row ------------------------------------------------------
                const int guess = last_index < cfg.n_alphas ? last_index : 0;
                const int mine = gl < cfg.n_alphas ? gl : cfg.n_alphas - 1;
                const float alpha = cfg.alphas[mine];
                float J, rmax;
                const int my_slot = gl - guess + kUserGroupSlotOfGuess;     // (lanes beyond the step sizes repeat the last one: they store nothing)
                const bool stores = my_slot >= 0 && my_slot < kUserGroupSlots && gl < cfg.n_alphas && live;
                const int ms = stores ? my_slot : kUserGroupSlotOfGuess;
                user_group_rollout<N, M, false>(pr, low, high, T, alpha, xhat, uhat, kg, stores, slot_x(ms), slot_u(ms), slot_c(ms), J, rmax);
                const float delta_J = -alpha * (dV1 + alpha * 0.0f);                                     // :339 (dV2 == 0)
                const float dcost = J_hat - J;
                const float z = (delta_J > 0.0f) ? dcost / delta_J : ((dcost > 0.0f) ? 1.0f : ((dcost < 0.0f) ? -1.0f : 0.0f));   // :342-346
                const unsigned pass = (unsigned)((__ballot(gl < cfg.n_alphas && z >= cfg.c1) >> row0) & (G == 32 ? 0xFFFFFFFFull : 0xFFFFull));
                const bool accept = pass != 0u;
                const int chosen = accept ? __builtin_ctz(pass) : cfg.n_alphas - 1;
                const float J_chosen = __shfl(J, row0 + chosen, kWave);
                const float residual = __shfl(rmax, row0 + chosen, kWave);
                wsync();
                int from = chosen - guess + kUserGroupSlotOfGuess;          // the slot the chosen step size's candidate is in
                if (from < 0 || from >= kUserGroupSlots) {                    // (row-uniform) not stored: once more, the whole row on that step size, lane 0 stores
                    float J2, r2;
                    user_group_rollout<N, M, false>(pr, low, high, T, cfg.alphas[chosen], xhat, uhat, kg, leader && live, xc, uc, cc, J2, r2);
                    wsync();
                    from = kUserGroupSlotOfGuess;
                }
                if (accept) last_index = chosen;
                const bool small_step = residual < cfg.atol;                  // :253-257 (taken even if rejected)
                if (leader && live)
                    trace_write(a.trace, b, iteration + attempts, iteration, mu, delta, J_hat, g_norm, chosen, cfg.alphas[chosen], J_chosen, accept ? 1 : 0,
                                residual, 0);
                if (small_step || accept) {             // the candidate becomes the nominal trajectory
                    if (live) {
                        const float *xf = slot_x(from), *uf = slot_u(from), *cf = slot_c(from);
                        for (int idx = gl; idx < (T + 1) * N; idx += G) xhat[idx] = xf[idx];
                        for (int idx = gl; idx < T * M; idx += G) uhat[idx] = uf[idx];
                        for (int idx = gl; idx <= T; idx += G) chat[idx] = cf[idx];
                    }
                    wsync();
                }
                if (small_step) {
                    converged = true;
                } else if (accept) {                                           // :259-266
                    delta = fminf(1.0f / cfg.delta_0, delta / cfg.delta_0);
                    mu = (mu * delta > cfg.mu_min) ? mu * delta : 0.0f;
                    if (++iteration >= cfg.max_iterations) finished = true;
                } else {                                                       // :267-270
                    delta = fmaxf(cfg.delta_0, delta * cfg.delta_0);
                    mu = fmaxf(cfg.mu_min, mu * delta);
                    if (++attempts >= cfg.max_attempts || !(mu < 1e30f)) give_up = true;
                }
            }
            if (converged || give_up) finished = true;
        }
    }
    if (give_up) status |= TFMPC_ST_MAX_ATTEMPTS;
    if (iteration >= cfg.max_iterations) iteration = cfg.max_iterations - 1;   // python's loop variable after exhaustion
    if (leader && live) {
        const float c0 = chat[T];
        if (!(c0 == c0)) status |= TFMPC_ST_NAN;
        a.iterations[b] = iteration;
        a.status[b] = status;
    }
}

}  // namespace tfmpc
