// ilqr_adjoint.hip -- iLQR.solve (tfmpc/solvers/ilqr.py:214-355) for the two reference envs whose
// costs are piecewise linear, HVAC (tfmpc/envs/hvac/__init__.py) and Reservoir
// (tfmpc/envs/reservoir/__init__.py), at n = m <= 32: BASELINE.json configs[4] (n = 32)
// and the reference's own hvac6 / res4 configs.
//
// On these envs every cost Hessian is identically zero, V_xx stays exactly 0, the backward pass always
// takes the bang-bang branch (ilqr.py:140-141, K_t = 0) and what is left of an iteration is the costate
// recursion  Q_x = l_x + f_x^T V_x,  Q_u = l_u + f_u^T V_x,  k = Q_u >= 0 ? low - u : high - u,  V_x <- Q_x
// plus open-loop rollouts u = clip(u_hat + alpha k).  That is mat-vec work, bound by
// instruction issue and LDS latency, so this kernel keeps the WHOLE env in registers:
//   * one wavefront = one instance; lane i owns state / action i for costs, gradients and the costate,
//     lane pair (2r, 2r+1) owns the two halves of row r of the coupling matrix for the transition;
//   * every parameter a lane needs (its matrix row / column, capacities, bounds ...) is loaded ONCE into
//     VGPRs; LDS (0.8 KB per wave) only carries the vectors lanes exchange: x (twice, so the rotated
//     column walk of the transition is an affine address), u and V_x;
//   * the nominal trajectory, the gains k_t and the candidate trajectory stream through the HBM workspace
//     one step ahead of their use;
//   * f_x, f_u are never formed: each lane multiplies its column of f_x^T on the fly.
// Arithmetic (operation order, fma placement, reduction trees) is that of the generic wave kernel
// (envs.h, ilqr_core.h): the two are bit-identical, which is how this kernel is tested.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <cstdlib>
#include <cstring>

#include "../../include/tfmpc_hip.h"
#include "ilqr_adjoint.h"
#include "options.h"
#include "trig.h"
#include "wave_ops.h"

namespace tfmpc {

namespace {

constexpr int kMaxN = 32, kHalf = 16;     // a lane pair splits a row of <= 32 into two runs of <= 16

__device__ __forceinline__ float sgnf(float y) { return (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f); }

// LDS slice of one wave (floats): x ping, x pong (each stored twice: [0,n) and [n,2n)), u, V_x
struct Slots {
    float *xa, *xb, *u, *vx;
};
constexpr int kXld = 2 * kMaxN + 4;       // doubled state vector + slack for the zero-coefficient tail reads
constexpr int kLdsFloats = 2 * kXld + kMaxN + kMaxN;

// A copy of the lane index the optimiser cannot see through: comparisons against it are redone where
// they are used instead of being hoisted out of the time loop as 32 loop-invariant lane masks (which
// would not fit the scalar register file and spill).
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// Reductions over a group of GS = 16 / 32 / 64 adjacent lanes (one instance per group).  Every value these kernels
// reduce lives in the first 16 lanes of its group, so the row tree of wave_sum / wave_max gives the same bits at any GS.
template <int GS>
__device__ __forceinline__ float group_sum(float v)
{
    if (GS == kWave) return wave_sum(v);
    v += dpp_move<kDppQuadXor1>(0.0f, v);
    v += dpp_move<kDppQuadXor2>(0.0f, v);
    v += dpp_move<kDppRowHalfMirror>(0.0f, v);
    v += dpp_move<kDppRowMirror>(0.0f, v);                 // every lane: sum of its row of 16
    if (GS == 32) v += __shfl_xor(v, 16, kWave);
    return v;
}
template <int GS>
__device__ __forceinline__ float group_max(float v)      // non-negative, non-NaN inputs (like wave_max)
{
    if (GS == kWave) return wave_max(v);
    v = fmaxf(v, dpp_move<kDppQuadXor1>(0.0f, v));
    v = fmaxf(v, dpp_move<kDppQuadXor2>(0.0f, v));
    v = fmaxf(v, dpp_move<kDppRowHalfMirror>(0.0f, v));
    v = fmaxf(v, dpp_move<kDppRowMirror>(0.0f, v));
    if (GS == 32) v = fmaxf(v, __shfl_xor(v, 16, kWave));
    return v;
}

// ---------------------------------------------------------------------------------- HVAC ----
// SMALL: n <= 16 -- loops stop at the first all-zero group (wave-uniform tests) and half the matrix registers; the
// n > 16 variant carries none of those tests in its time loops.
template <int KIND, bool SMALL, int GS = kWave> struct Lean;      // GS: lanes per instance (lane index = gl)

template <bool SMALL, int GS> struct Lean<TFMPC_ENV_HVAC, SMALL, GS> {
    int gl;                                // lane within the instance's group
    static constexpr float CAP_AIR = 1.006f, COST_AIR = 1.0f, TEMP_AIR = 40.0f, TIME_DELTA = 1.0f;
    static constexpr float PENALTY = 20000.0f, SET_POINT_PENALTY = 10.0f;
    int n, it, part, off, tail, per;       // transition: row it = lane / parts, run [j0, j0 + cnt), off = it + j0
    static constexpr int kRun = SMALL ? 8 : kHalf, kCols = SMALL ? kHalf : kMaxN;   // longest run of a lane; columns
    float Grow[kRun];                      // G[it][(j0 + j + it) mod n] for j < tail = cnt & ~3, else 0
    float Gtail[3];                        // the cnt & 3 elements after them (else 0)
    float t_out, t_hall, k_out, k_hall, rcap, am_t;      // row it
    float lo, hi, mid, am;                 // lane i < n: bounds, their midpoint (lo + hi) / 2, air_max
    float coef[kCols];                     // lane i < n: dtc[kk] * G[kk][i], with 0 at kk == i (added separately)
    float Gii, gsum, dtc_i, k_out_i, k_hall_i;           // lane i < n: the diagonal of f_x
    float dtc_a, am_a;                     // lane n + a: the diagonal of f_u

    __device__ void load(const TfmpcEnv &g, int b)
    {
        const int lane = gl = lane_id() % GS;
        n = g.n;
        auto P = [&](int i) { return g.p[i] + (size_t)b * g.stride[i]; };
        const float *pt_out = P(0), *pt_hall = P(1), *plo = P(2), *phi = P(3), *pk_out = P(4), *pk_hall = P(5), *pcap = P(6),
                    *pam = P(7), *G = P(8);
        it = lane >> 1; part = lane & 1;
        per = (n + 1) / 2;
        const int j0 = part * per, j1 = (j0 + per < n) ? j0 + per : n;
        const int cnt = (it < n && j1 > j0) ? j1 - j0 : 0;
        off = (it < n) ? it + j0 : 0;
        tail = cnt & ~3;
        auto Gat = [&](int j) {
            int c = j0 + j + it;
            c -= (c >= n) ? n : 0;
            return G[it * n + c];
        };
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            if ((j & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            Grow[j] = (j < tail) ? Gat(j) : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) Gtail[q] = (tail + q < cnt) ? Gat(tail + q) : 0.0f;
        const bool row = it < n;
        t_out = row ? pt_out[it] : 0.0f; t_hall = row ? pt_hall[it] : 0.0f; k_out = row ? pk_out[it] : 0.0f;
        k_hall = row ? pk_hall[it] : 0.0f; rcap = row ? TIME_DELTA / pcap[it] : 0.0f; am_t = row ? pam[it] : 0.0f;
        const int i = lane;
        const bool st = i < n;
        lo = st ? plo[i] : 0.0f; hi = st ? phi[i] : 0.0f; am = st ? pam[i] : 0.0f;
        mid = (lo + hi) / 2;
        dtc_i = st ? TIME_DELTA / pcap[i] : 0.0f; k_out_i = st ? pk_out[i] : 0.0f; k_hall_i = st ? pk_hall[i] : 0.0f;
        Gii = st ? G[i * n + i] : 0.0f;
        float gs = 0.0f;
        if (st) for (int k = 0; k < n; ++k) gs += G[i * n + k];
        gsum = gs;
#pragma unroll
        for (int kk = 0; kk < kCols; ++kk) {
            if ((kk & 3) == 0) __builtin_amdgcn_sched_barrier(0);     // one-time code: keep few loads in flight
            coef[kk] = (st && kk < n && kk != i) ? (TIME_DELTA / pcap[kk]) * G[kk * n + i] : 0.0f;
        }
        const int a = lane - n;
        const bool ac = a >= 0 && a < n;
        dtc_a = ac ? TIME_DELTA / pcap[a] : 0.0f;
        am_a = ac ? pam[a] : 0.0f;
    }
    __device__ __forceinline__ float penalties(float x) const
    {
        const float oob = PENALTY * (fmaxf(0.0f, lo - x) + fmaxf(0.0f, x - hi));      // hvac :97-100
        const float sp = SET_POINT_PENALTY * fabsf(mid - x);                 // :101-105
        return oob + sp;
    }
    // stage cost (:91-110) and final cost (:112-129); x, u in LDS
    __device__ float cost(const float *x, const float *u) const
    {
        float part_ = 0.0f;
        if (gl < n) part_ += COST_AIR * (u[gl] * am) + penalties(x[gl]);
        return group_sum<GS>(part_);
    }
    __device__ float final_cost(const float *x) const
    {
        float part_ = 0.0f;
        if (gl < n) part_ += penalties(x[gl]);
        return group_sum<GS>(part_);
    }
    __device__ __forceinline__ float grad_x(float x) const
    {
        return PENALTY * (-(lo > x ? 1.0f : 0.0f) + (x > hi ? 1.0f : 0.0f)) - SET_POINT_PENALTY * sgnf(mid - x);
    }
    // x' (:69-89): x is the doubled state vector x2[0..2n), so the rotated walk (j0 + j + it) mod n is
    // x2[off + j]
    __device__ void transition(const float *x2, const float *u, float *xn2) const
    {
        // Four partial sums over the run in steps of 4, then the (up to 3) left-over elements into the
        // first one -- the generic kernel's order; elements beyond the run carry a zero coefficient
        // (an exact no-op) instead of a branch.
        const float xi = x2[(it < n) ? it : 0];
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        const float *xr = x2 + off;
#pragma unroll
        for (int gq = 0; gq < kRun / 4; ++gq) {
            if (gq == 2) __builtin_amdgcn_sched_barrier(0);
            if (SMALL && 4 * gq + 4 > per) break;            // wave-uniform: no run reaches this group of four
            s0 = fmaf(-Grow[4 * gq], xi - xr[4 * gq], s0);
            s1 = fmaf(-Grow[4 * gq + 1], xi - xr[4 * gq + 1], s1);
            s2 = fmaf(-Grow[4 * gq + 2], xi - xr[4 * gq + 2], s2);
            s3 = fmaf(-Grow[4 * gq + 3], xi - xr[4 * gq + 3], s3);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) s0 = fmaf(-Gtail[q], xi - xr[tail + q], s0);
        float between = (s0 + s1) + (s2 + s3);
        between += quad_xor1(between);
        if (it < n && part == 0) {
            const float air = u[it] * am_t;                                               // :72
            const float heating = air * CAP_AIR * (TEMP_AIR - xi);                        // :74
            const float outside = k_out * (t_out - xi);                                   // :143-144
            const float hall = k_hall * (t_hall - xi);                                    // :148-149
            const float v = xi + rcap * (heating + between + outside + hall);             // :80-88
            xn2[it] = v;
            xn2[n + it] = v;
        }
    }
    // Q_x[i] (lanes < n) and Q_u[a] (lanes n + a) of one backward step; xh, uh, vx in LDS
    __device__ __forceinline__ float adjoint(const float *xh, const float *uh, const float *vx) const
    {
        const int lane = gl;
        float acc = 0.0f;
        if (lane < n) {
            const float diag = 1.0f + dtc_i * (Gii - uh[lane] * am * CAP_AIR - gsum - k_out_i - k_hall_i);
            // diagonal term first, then the column by ascending row: its own diagonal slot holds 0, an
            // exact no-op (as do the rows >= n: 0 * 0) -- the order of ilqr_core.h backward_pass
            acc = fmaf(diag, vx[lane], grad_x(xh[lane]));
#pragma unroll
            for (int kk = 0; kk < kCols; ++kk) {
                if ((kk & 7) == 0) __builtin_amdgcn_sched_barrier(0);     // 8 V_x values in flight, not 32
                if (SMALL && (kk & 7) == 0 && kk >= n) break;             // wave-uniform: the rest are 0 * 0
                acc = fmaf(coef[kk], vx[kk], acc);
            }
        } else if (lane < 2 * n) {
            const int a = lane - n;
            const float d = dtc_a * am_a * CAP_AIR * (TEMP_AIR - xh[a]);
            acc = fmaf(d, vx[a], COST_AIR * am_a);
        }
        return acc;
    }
};

// ----------------------------------------------------------------------------- RESERVOIR ----
template <bool SMALL, int GS> struct Lean<TFMPC_ENV_RESERVOIR, SMALL, GS> {
    int gl;                                // lane within the instance's group
    int n, it, part, j0, tail;
    static constexpr int kRun = SMALL ? 8 : kHalf, kCols = SMALL ? kHalf : kMaxN;   // longest run of a lane; columns
    float Dcol[kRun];                      // transition: D[j0 + j][it] for j < tail = cnt & ~3, else 0
    float Dtail[3];                        // the cnt & 3 elements after them (else 0)
    float cap_t, rain_t;                   // row it
    float cap, lo, hi, mid, LP, HP, SP;    // lane i < n (mid = (lo + hi) / 2)
    float Drow[kCols];                     // lanes i < n and n + a: D[i][kk] (row i = lane mod n), 0 at kk == i
    float Dii;                             // D[i][i], added separately

    __device__ void load(const TfmpcEnv &g, int b)
    {
        const int lane = gl = lane_id() % GS;
        n = g.n;
        auto P = [&](int i) { return g.p[i] + (size_t)b * g.stride[i]; };
        const float *pcap = P(0), *plo = P(1), *phi = P(2), *plp = P(3), *php = P(4), *psp = P(5), *prain = P(6), *D = P(7);
        it = lane >> 1; part = lane & 1;
        const int per = (n + 1) / 2;
        j0 = part * per;
        const int j1 = (j0 + per < n) ? j0 + per : n;
        const int cnt = (it < n && j1 > j0) ? j1 - j0 : 0;
        if (it >= n) j0 = 0;
        tail = cnt & ~3;
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            if ((j & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            Dcol[j] = (j < tail) ? D[(j0 + j) * n + it] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) Dtail[q] = (tail + q < cnt) ? D[(j0 + tail + q) * n + it] : 0.0f;
        cap_t = (it < n) ? pcap[it] : 1.0f;
        rain_t = (it < n) ? prain[it] : 0.0f;
        const int i = lane;
        const bool st = i < n;
        cap = st ? pcap[i] : 1.0f; lo = st ? plo[i] : 0.0f; hi = st ? phi[i] : 0.0f;
        mid = (lo + hi) / 2.0f;
        LP = st ? -plp[i] : 0.0f; HP = st ? -php[i] : 0.0f; SP = st ? -psp[i] : 0.0f;
        const int rrow = (lane < n) ? lane : lane - n;
        const bool rv = rrow >= 0 && rrow < n && lane < 2 * n;
        Dii = rv ? D[rrow * n + rrow] : 0.0f;
#pragma unroll
        for (int kk = 0; kk < kCols; ++kk) {
            if ((kk & 3) == 0) __builtin_amdgcn_sched_barrier(0);     // one-time code: keep few loads in flight
            Drow[kk] = (rv && kk < n && kk != rrow) ? D[rrow * n + kk] : 0.0f;
        }
    }
    __device__ float cost(const float *x, const float *) const                            // reservoir :63-79
    {
        float part_ = 0.0f;
        if (gl < n) {
            const float xv = x[gl];
            const float c1 = LP * fmaxf(0.0f, lo - xv);
            const float c2 = HP * fmaxf(0.0f, xv - hi);
            const float c3 = SP * fabsf(mid - xv);
            part_ += c1 + c2 + c3;
        }
        return group_sum<GS>(part_);
    }
    __device__ float final_cost(const float *x) const { return cost(x, nullptr); }         // :81-83
    __device__ __forceinline__ float grad_x(float x) const
    {
        return -LP * (lo > x ? 1.0f : 0.0f) + HP * (x > hi ? 1.0f : 0.0f) - SP * sgnf(mid - x);
    }
    __device__ void transition(const float *x2, const float *u, float *xn2) const          // :47-61, :85-105
    {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        const float *xr = x2 + j0, *ur = u + j0;
#pragma unroll
        for (int gq = 0; gq < kRun / 4; ++gq) {
            if (gq == 2) __builtin_amdgcn_sched_barrier(0);
            if (SMALL && 4 * gq + 4 > (n + 1) / 2) break;    // wave-uniform: no run reaches this group of four
            s0 = fmaf(Dcol[4 * gq], ur[4 * gq] * xr[4 * gq], s0);
            s1 = fmaf(Dcol[4 * gq + 1], ur[4 * gq + 1] * xr[4 * gq + 1], s1);
            s2 = fmaf(Dcol[4 * gq + 2], ur[4 * gq + 2] * xr[4 * gq + 2], s2);
            s3 = fmaf(Dcol[4 * gq + 3], ur[4 * gq + 3] * xr[4 * gq + 3], s3);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) s0 = fmaf(Dtail[q], ur[tail + q] * xr[tail + q], s0);
        float inflow = (s0 + s1) + (s2 + s3);
        inflow += quad_xor1(inflow);
        if (it < n && part == 0) {
            const float xi = x2[it];
            const float vaporated = 0.5f * sin_f32(xi * (1.0f / cap_t)) * xi;                          // :87
            const float v = xi + rain_t + inflow - vaporated - u[it] * xi;                 // :56-60
            xn2[it] = v;
            xn2[n + it] = v;
        }
    }
    __device__ __forceinline__ float adjoint(const float *xh, const float *uh, const float *vx) const
    {
        const int lane = gl;
        float acc = 0.0f;
        if (lane < n) {
            const float uj = uh[lane];
            const float r = xh[lane] * (1.0f / cap);
            float sr, cr;
            sincos_f32(r, sr, cr);
            const float diag_extra = 1.0f - 0.5f * (cr * r + sr) - uj;
            // diagonal term first, then the row by ascending column with a zero in the diagonal slot
            acc = fmaf(Dii * uj + diag_extra, vx[lane], grad_x(xh[lane]));
#pragma unroll
            for (int kk = 0; kk < kCols; ++kk) {                  // kk >= n: 0 * 0
                if ((kk & 7) == 0) __builtin_amdgcn_sched_barrier(0);     // 8 V_x values in flight, not 32
                if (SMALL && (kk & 7) == 0 && kk >= n) break;
                acc = fmaf(Drow[kk] * uj, vx[kk], acc);
            }
        } else if (lane < 2 * n) {
            const int a_ = lane - n;
            const float xa = xh[a_];
            acc = fmaf(Dii * xa - xa, vx[a_], 0.0f);
#pragma unroll
            for (int kk = 0; kk < kCols; ++kk) {
                if ((kk & 7) == 0) __builtin_amdgcn_sched_barrier(0);
                if (SMALL && (kk & 7) == 0 && kk >= n) break;
                acc = fmaf(Drow[kk] * xa, vx[kk], acc);
            }
        }
        return acc;
    }
};

struct BackwardOut { float J, dV1, g_norm; };

// 4 waves per SIMD (<= 128 VGPR, ~18 rarely used values in scratch): measured best of 3 / 4 / 5 / 6 on
// cfg5 (the kernel is latency-bound; below 128 VGPR the spills reach the time loops); the n <= 16 variant keeps half the
// matrix registers and runs 6 waves per SIMD (measured best of 4 / 5 / 6 / 8 on hvac6 and res4)
template <int KIND, bool SMALL>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(SMALL ? 6 : 4, SMALL ? 6 : 4))) void ilqr_adjoint_solve_kernel(TfmpcEnv genv, TfmpcIlqrConfig cfg, AdjointSolveArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    const int b = blockIdx.x, lane = lane_id(), n = genv.n, m = n, T = a.T;
    Lean<KIND, SMALL> env;
    env.load(genv, b);
    float *xa = lds, *xb = lds + kXld, *ul = lds + 2 * kXld, *vx = lds + 2 * kXld + kMaxN;
    for (int idx = lane; idx < kLdsFloats; idx += kWave) lds[idx] = 0.0f;      // entries >= n stay 0 (zero-coefficient reads)
    wsync();
    const float low = (lane < m) ? genv.low[lane] : 0.0f, high = (lane < m) ? genv.high[lane] : 0.0f;
    const int al = lane - n;                             // action index of the costate's Q_u lanes
    const float low_a = (al >= 0 && al < m) ? genv.low[al] : 0.0f, high_a = (al >= 0 && al < m) ? genv.high[al] : 0.0f;

    float *xhat = a.states + (size_t)b * (T + 1) * n, *uhat = a.actions + (size_t)b * T * m, *chat = a.costs + (size_t)b * (T + 1);
    float *kg = a.wsk + (size_t)b * T * m;
    float *xc = a.wsx + (size_t)b * (T + 1) * n, *uc = a.wsu + (size_t)b * T * m, *cc = a.wsc + (size_t)b * (T + 1);

    // one rollout: u_t from `next_u(t)` (called one step ahead by lanes < m), trajectory to (xs, us, cs)
    auto rollout = [&](auto next_u, float *xs, float *us, float *cs, float &J_out) {
        float *xcur = xa, *xnext = xb;
        if (lane < n) { const float x = xhat == xs ? a.x0[(size_t)b * n + lane] : xhat[lane]; xcur[lane] = x; xcur[n + lane] = x; xs[lane] = x; }
        float J = 0.0f;
        float u_n = (lane < m && T > 0) ? next_u(0) : 0.0f;
        for (int t = 0; t < T; ++t) {
            const float u_c = u_n;
            if (lane < m) { ul[lane] = u_c; (us + (size_t)t * m)[lane] = u_c; }
            if (lane < m && t + 1 < T) u_n = next_u(t + 1);
            wsync();
            const float c = env.cost(xcur, ul);
            env.transition(xcur, ul, xnext);
            J += c;
            if (lane == 0) cs[t] = c;
            wsync();
            if (lane < n) (xs + (size_t)(t + 1) * n)[lane] = xnext[lane];
            float *tmp = xcur; xcur = xnext; xnext = tmp;
        }
        wsync();
        const float fc = env.final_cost(xcur);
        if (lane == 0) cs[T] = fc;
        J_out = J + fc;
        wsync();
    };

    // start (ilqr.py:218, :53-82): nominal trajectory from the injected actions
    {
        const float *u0 = a.u_init + (size_t)b * T * m;
        float J;
        rollout([&](int t) { return (u0 + (size_t)t * m)[lane]; }, xhat, uhat, chat, J);
    }

    float mu = 0.0f, delta = 1.0f;
    int status = 0, attempts = 0, iteration = 0;
    bool converged = false, give_up = false;
    for (iteration = 0; iteration < cfg.max_iterations; ++iteration) {
        for (;;) {
            // ---- backward (ilqr.py:94-172 on the bang-bang branch): costate recursion ------------
            BackwardOut r{0.0f, 0.0f, 0.0f};
            {
                if (lane < n) { const float x = (xhat + (size_t)T * n)[lane]; xa[lane] = x; xa[n + lane] = x; }
                wsync();
                if (lane < n) vx[lane] = env.grad_x(xa[lane]);                 // V_x = l_x^f
                r.J = env.final_cost(xa);
                float gsum = 0.0f;
                float x_n = 0.0f, u_n = 0.0f;
                if (T > 0) {
                    if (lane < n) x_n = (xhat + (size_t)(T - 1) * n)[lane];
                    else if (lane < n + m) u_n = (uhat + (size_t)(T - 1) * m)[al];
                }
                wsync();
                for (int t = T - 1; t >= 0; --t) {
                    if (lane < n) { xa[lane] = x_n; xa[n + lane] = x_n; }
                    else if (lane < n + m) ul[al] = u_n;
                    const float uh_a = u_n;                                     // lanes n + a keep u_hat[a]
                    if (t > 0) {
                        if (lane < n) x_n = (xhat + (size_t)(t - 1) * n)[lane];
                        else if (lane < n + m) u_n = (uhat + (size_t)(t - 1) * m)[al];
                    }
                    wsync();
                    const float l = env.cost(xa, ul);
                    const float acc = env.adjoint(xa, ul, vx);
                    float p1 = 0.0f, gmax = 0.0f;
                    if (lane >= n && lane < n + m) {
                        const float kt = (acc >= 0.0f) ? (low_a - uh_a) : (high_a - uh_a);       // :140-141
                        (kg + (size_t)t * m)[al] = kt;
                        p1 = fmaf(kt, acc, p1);
                        gmax = fmaxf(gmax, fabsf(kt) / (fabsf(uh_a) + 1.0f));
                    }
                    r.J += l;
                    r.dV1 += wave_sum(p1);
                    gsum += wave_max(gmax);
                    wsync();
                    if (lane < n) vx[lane] = acc;                               // V_x <- Q_x
                    wsync();
                }
                r.g_norm = T > 0 ? gsum / (float)T : 0.0f;
            }
            if (r.g_norm < cfg.atol) { converged = true; break; }               // :243-248
            wsync();
            // ---- line search (ilqr.py:317-355; rollouts :174-212 with K == 0) --------------------
            bool accept = false;
            float residual = 0.0f;
            for (int ai = 0; ai < cfg.n_alphas; ++ai) {
                const float alpha = cfg.alphas[ai];
                float J, rmax = 0.0f;
                rollout([&](int t) {
                            const float du = alpha * (kg + (size_t)t * m)[lane];                   // :193-194
                            rmax = fmaxf(rmax, fabsf(du));
                            return fminf(fmaxf((uhat + (size_t)t * m)[lane] + du, low), high);     // :196-197
                        },
                        xc, uc, cc, J);
                residual = wave_max(rmax);                                      // :206
                const float delta_J = -alpha * (r.dV1 + alpha * 0.0f);          // :339 (dV2 == 0 here)
                const float dcost = r.J - J;
                const float z = (delta_J > 0.0f) ? dcost / delta_J : sgnf(dcost);   // :342-346
                if (z >= cfg.c1) { accept = true; break; }                      // :351-353
            }
            const bool small_step = residual < cfg.atol;                       // :253-257
            if (small_step || accept) {
                for (int idx = lane; idx < (T + 1) * n; idx += kWave) xhat[idx] = xc[idx];
                for (int idx = lane; idx < T * m; idx += kWave) uhat[idx] = uc[idx];
                for (int idx = lane; idx <= T; idx += kWave) chat[idx] = cc[idx];
                wsync();
            }
            if (small_step) { converged = true; break; }
            if (accept) {                                                       // :259-266
                delta = fminf(1.0f / cfg.delta_0, delta / cfg.delta_0);
                mu = (mu * delta > cfg.mu_min) ? mu * delta : 0.0f;
                break;
            }
            delta = fmaxf(cfg.delta_0, delta * cfg.delta_0);                    // :267-270
            mu = fmaxf(cfg.mu_min, mu * delta);
            if (++attempts >= cfg.max_attempts || !(mu < 1e30f)) { give_up = true; break; }
        }
        if (converged || give_up) break;                                        // :276-277
    }
    if (give_up) status |= TFMPC_ST_MAX_ATTEMPTS;
    if (iteration >= cfg.max_iterations) iteration = cfg.max_iterations - 1;
    if (lane == 0) {
        const float c0 = chat[T];
        if (!(c0 == c0)) status |= TFMPC_ST_NAN;
        a.iterations[b] = iteration;
        a.status[b] = status;
    }
}

// ---- several instances per wavefront (n <= 8) ------------------------------------------------------------------
// A 6-room HVAC instance keeps 24 of the 64 lanes busy, a 4-reservoir instance 8.  Here a wave carries 64 / GS
// instances, one per group of GS = 16 or 32 lanes, in lockstep: every tick all groups run one costate recursion, then
// line-search rounds in which every group still searching rolls out ITS next step size; groups that are finished, or
// have accepted, keep executing (their stores are masked) until the last group of the wave is done.  Per group the
// state machine and the arithmetic are those of ilqr_adjoint_solve_kernel (so of the generic wave kernel: bit-identical,
// which is how this is tested); scalars that were wave-uniform there are group-uniform vector values here.
// 4 waves per SIMD: measured best of 3 / 4 / 5 / 6 / 8 on hvac6 and res4 (more group state, fewer spills).
template <int KIND, int GS>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(4, 4))) void ilqr_adjoint_group_kernel(TfmpcEnv genv, TfmpcIlqrConfig cfg, AdjointSolveArgs a)
{
    constexpr int G = kWave / GS;
    __shared__ __attribute__((aligned(16))) float lds[G * kLdsFloats];
    const int wl = lane_id(), grp = wl / GS, lane = wl % GS, n = genv.n, m = n, T = a.T;
    const int b_raw = blockIdx.x * G + grp;
    const bool live = b_raw < a.B;                       // the last wave may carry empty groups: they compute on the
    const int b = live ? b_raw : a.B - 1;                // last instance's data and store nothing
    Lean<KIND, true, GS> env;
    env.load(genv, b);
    float *slice = lds + grp * kLdsFloats;
    float *xa = slice, *xb = slice + kXld, *ul = slice + 2 * kXld, *vx = slice + 2 * kXld + kMaxN;
    for (int idx = wl; idx < G * kLdsFloats; idx += kWave) lds[idx] = 0.0f;
    wsync();
    const float low = (lane < m) ? genv.low[lane] : 0.0f, high = (lane < m) ? genv.high[lane] : 0.0f;
    const int al = lane - n;
    const float low_a = (al >= 0 && al < m) ? genv.low[al] : 0.0f, high_a = (al >= 0 && al < m) ? genv.high[al] : 0.0f;

    float *xhat = a.states + (size_t)b * (T + 1) * n, *uhat = a.actions + (size_t)b * T * m, *chat = a.costs + (size_t)b * (T + 1);
    float *kg = a.wsk + (size_t)b * T * m;
    float *xc = a.wsx + (size_t)b * (T + 1) * n, *uc = a.wsu + (size_t)b * T * m, *cc = a.wsc + (size_t)b * (T + 1);

    // one rollout of every group; `keep` (group-uniform) masks the stores of the trajectory
    auto rollout = [&](auto next_u, bool from_x0, bool keep, float *xs, float *us, float *cs, float &J_out) {
        float *xcur = xa, *xnext = xb;
        if (lane < n) {
            const float x = from_x0 ? a.x0[(size_t)b * n + lane] : xhat[lane];
            xcur[lane] = x; xcur[n + lane] = x;
            if (keep) xs[lane] = x;
        }
        float J = 0.0f;
        float u_n = (lane < m && T > 0) ? next_u(0) : 0.0f;
        for (int t = 0; t < T; ++t) {
            const float u_c = u_n;
            if (lane < m) { ul[lane] = u_c; if (keep) (us + (size_t)t * m)[lane] = u_c; }
            if (lane < m && t + 1 < T) u_n = next_u(t + 1);
            wsync();
            const float c = env.cost(xcur, ul);
            env.transition(xcur, ul, xnext);
            J += c;
            if (lane == 0 && keep) cs[t] = c;
            wsync();
            if (lane < n && keep) (xs + (size_t)(t + 1) * n)[lane] = xnext[lane];
            float *tmp = xcur; xcur = xnext; xnext = tmp;
        }
        wsync();
        const float fc = env.final_cost(xcur);
        if (lane == 0 && keep) cs[T] = fc;
        J_out = J + fc;
        wsync();
    };

    {
        const float *u0 = a.u_init + (size_t)b * T * m;
        float J;
        rollout([&](int t) { return (u0 + (size_t)t * m)[lane]; }, true, live, xhat, uhat, chat, J);
    }

    float mu = 0.0f, delta = 1.0f;
    int status = 0, attempts = 0, iteration = 0;
    bool done = !live || cfg.max_iterations <= 0;
    while (__any(!done)) {
        // ---- backward (ilqr.py:94-172 on the bang-bang branch): costate recursion, all groups ------------
        BackwardOut r{0.0f, 0.0f, 0.0f};
        {
            if (lane < n) { const float x = (xhat + (size_t)T * n)[lane]; xa[lane] = x; xa[n + lane] = x; }
            wsync();
            if (lane < n) vx[lane] = env.grad_x(xa[lane]);
            r.J = env.final_cost(xa);
            float gsum = 0.0f;
            float x_n = 0.0f, u_n = 0.0f;
            if (T > 0) {
                if (lane < n) x_n = (xhat + (size_t)(T - 1) * n)[lane];
                else if (lane < n + m) u_n = (uhat + (size_t)(T - 1) * m)[al];
            }
            wsync();
            for (int t = T - 1; t >= 0; --t) {
                if (lane < n) { xa[lane] = x_n; xa[n + lane] = x_n; }
                else if (lane < n + m) ul[al] = u_n;
                const float uh_a = u_n;
                if (t > 0) {
                    if (lane < n) x_n = (xhat + (size_t)(t - 1) * n)[lane];
                    else if (lane < n + m) u_n = (uhat + (size_t)(t - 1) * m)[al];
                }
                wsync();
                const float l = env.cost(xa, ul);
                const float acc = env.adjoint(xa, ul, vx);
                float p1 = 0.0f, gmax = 0.0f;
                if (lane >= n && lane < n + m) {
                    const float kt = (acc >= 0.0f) ? (low_a - uh_a) : (high_a - uh_a);
                    if (!done) (kg + (size_t)t * m)[al] = kt;
                    p1 = fmaf(kt, acc, p1);
                    gmax = fmaxf(gmax, fabsf(kt) / (fabsf(uh_a) + 1.0f));
                }
                r.J += l;
                r.dV1 += group_sum<GS>(p1);
                gsum += group_max<GS>(gmax);
                wsync();
                if (lane < n) vx[lane] = acc;
                wsync();
            }
            r.g_norm = T > 0 ? gsum / (float)T : 0.0f;
        }
        const bool converged_g = !done && r.g_norm < cfg.atol;                 // :243-248
        wsync();
        // ---- line search rounds (ilqr.py:317-355): every searching group tries its next step size ------
        const bool searching = !done && !converged_g;
        bool accept = false;
        float residual = 0.0f;
        for (int ai = 0; ai < cfg.n_alphas && __any(searching && !accept); ++ai) {
            const float alpha = cfg.alphas[ai];
            const bool trying = searching && !accept;
            float J, rmax = 0.0f;
            rollout([&](int t) {
                        const float du = alpha * (kg + (size_t)t * m)[lane];
                        rmax = fmaxf(rmax, fabsf(du));
                        return fminf(fmaxf((uhat + (size_t)t * m)[lane] + du, low), high);
                    },
                    false, trying, xc, uc, cc, J);
            const float res = group_max<GS>(rmax);
            const float delta_J = -alpha * (r.dV1 + alpha * 0.0f);
            const float dcost = r.J - J;
            const float z = (delta_J > 0.0f) ? dcost / delta_J : sgnf(dcost);
            if (trying) {
                residual = res;
                if (z >= cfg.c1) accept = true;
            }
        }
        const bool small_step = searching && residual < cfg.atol;            // :253-257
        const bool take = searching && (small_step || accept);
        {
            for (int idx = lane; idx < (T + 1) * n; idx += GS) { const float v = xc[idx]; if (take) xhat[idx] = v; }
            for (int idx = lane; idx < T * m; idx += GS) { const float v = uc[idx]; if (take) uhat[idx] = v; }
            for (int idx = lane; idx <= T; idx += GS) { const float v = cc[idx]; if (take) chat[idx] = v; }
            wsync();
        }
        if (converged_g || small_step) done = true;                            // converged
        else if (searching && accept) {                                        // :259-266
            delta = fminf(1.0f / cfg.delta_0, delta / cfg.delta_0);
            mu = (mu * delta > cfg.mu_min) ? mu * delta : 0.0f;
            if (++iteration >= cfg.max_iterations) done = true;
        } else if (searching) {                                                // :267-270
            delta = fmaxf(cfg.delta_0, delta * cfg.delta_0);
            mu = fmaxf(cfg.mu_min, mu * delta);
            if (++attempts >= cfg.max_attempts || !(mu < 1e30f)) { status |= TFMPC_ST_MAX_ATTEMPTS; done = true; }
        }
    }
    if (iteration >= cfg.max_iterations) iteration = cfg.max_iterations - 1;
    if (lane == 0 && live) {
        const float c0 = chat[T];
        if (!(c0 == c0)) status |= TFMPC_ST_NAN;
        a.iterations[b] = iteration;
        a.status[b] = status;
    }
}

}  // namespace

bool ilqr_adjoint_supported(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg)
{
    return (env.kind == TFMPC_ENV_HVAC || env.kind == TFMPC_ENV_RESERVOIR) && env.n == env.m && env.n >= 2 && env.n <= kMaxN &&
           env.bounded && !cfg.storage_bf16;
}

int ilqr_adjoint_launch(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, const AdjointSolveArgs &a, hipStream_t stream)
{
    const dim3 block(kWave);
    const bool small = env.n <= kHalf;
    // lanes an instance needs: two per row of the transition, and 2 n for Q_x, Q_u
    const int lanes = 2 * env.n;
    const bool packed = lanes <= 32 && !option_is(kOptIlqrKernel, "lean1");     // "lean1": one instance per wave (A/B, tests)
    if (packed && lanes <= 16) {
        const dim3 grid((a.B + 3) / 4);
        if (env.kind == TFMPC_ENV_HVAC) hipLaunchKernelGGL((ilqr_adjoint_group_kernel<TFMPC_ENV_HVAC, 16>), grid, block, 0, stream, env, cfg, a);
        else hipLaunchKernelGGL((ilqr_adjoint_group_kernel<TFMPC_ENV_RESERVOIR, 16>), grid, block, 0, stream, env, cfg, a);
    } else if (packed) {
        const dim3 grid((a.B + 1) / 2);
        if (env.kind == TFMPC_ENV_HVAC) hipLaunchKernelGGL((ilqr_adjoint_group_kernel<TFMPC_ENV_HVAC, 32>), grid, block, 0, stream, env, cfg, a);
        else hipLaunchKernelGGL((ilqr_adjoint_group_kernel<TFMPC_ENV_RESERVOIR, 32>), grid, block, 0, stream, env, cfg, a);
    } else {
        const dim3 grid(a.B);
        if (env.kind == TFMPC_ENV_HVAC) {
            if (small) hipLaunchKernelGGL((ilqr_adjoint_solve_kernel<TFMPC_ENV_HVAC, true>), grid, block, 0, stream, env, cfg, a);
            else hipLaunchKernelGGL((ilqr_adjoint_solve_kernel<TFMPC_ENV_HVAC, false>), grid, block, 0, stream, env, cfg, a);
        } else {
            if (small) hipLaunchKernelGGL((ilqr_adjoint_solve_kernel<TFMPC_ENV_RESERVOIR, true>), grid, block, 0, stream, env, cfg, a);
            else hipLaunchKernelGGL((ilqr_adjoint_solve_kernel<TFMPC_ENV_RESERVOIR, false>), grid, block, 0, stream, env, cfg, a);
        }
    }
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}

}  // namespace tfmpc
