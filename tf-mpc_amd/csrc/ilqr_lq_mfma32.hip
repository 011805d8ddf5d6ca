// ilqr_lq_mfma32.hip -- iLQR.solve (tfmpc/solvers/ilqr.py:214-355) on the matrix cores for the time-invariant LQ env
// with UNBOUNDED actions at shapes beyond the 16 x 8 tile, up to n = 32, m = 16: BASELINE configs[4] at its literal dims
// (state_dim 32, action_dim 16, horizon 100) driven through the iLQR API.  The large-tile twin of ilqr_lq_mfma.hip:
//
//   * backward pass (ilqr.py:94-172 at mu = 0) = the 2 x 2-tile bf16x3 Riccati sweep of lqr_mfma32x16.hip whose affine
//     column carries (l_z(t) + F^T V_x): V_xx' = Q_xx + Q_xu K, V_x' = Q_x + Q_xu k, dV2 = -dV1 / 2;
//   * the cost gradients l_z(t) = C_s z_t + c of the whole nominal trajectory are one C Z product on the f32 matrix cores
//     per iteration (16 timesteps per tile), the stage costs of every rollout another one;
//   * trajectories do NOT fit LDS at this shape (T = 100: 3 x 21 KB per wave): the nominal trajectory lives in the output
//     arrays or the workspace (swapped on accept, one copy-out at the end if needed), the gradients borrow the candidate
//     buffer during the sweep, the rollout runs in LDS chunks of 48 timesteps with bulk stores.
// What this kernel does not implement -- a non-PD Q_uu (needs mu > 0, ilqr.py:305-309) or a line search that rejects all
// 11 steps (ilqr.py:267-270) -- it reports with kIlqrRetryBit; the dispatcher runs the wave kernel over exactly those
// instances (as for ilqr_lq_mfma.hip).
#include <hip/hip_runtime.h>

#include "ilqr_lq_mfma.h"
#include "mfma_bf16x3.h"
#include "options.h"
#include "wave_ldlt.h"
#include "wave_ops.h"

namespace tfmpc {

namespace {

constexpr int N = 32, M = 16, D = 48;
using f32x4 = bf3::f32x4;
using namespace bf3;

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float readlane(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E;
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float sgn(float y) { return (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f); }

// per-wave LDS slice (floats), as lqr_mfma32x16.hip: the rollout's chunk buffer reuses the sweep's staging areas
constexpr int kMld = 64;
constexpr int kMs = 0;
constexpr int kKs = kMs + M * kMld;
constexpr int kkv = kKs + M * N;
constexpr int kVt = kkv + 16;
constexpr int kVld = 33;
constexpr int kZeros = kVt + N * kVld;
constexpr int kSweepFloats = kZeros + 16;
constexpr int kZld = 52;
constexpr int kTC = 48;
constexpr int kLdsFloats = (kSweepFloats > (kTC + 1) * kZld ? kSweepFloats : (kTC + 1) * kZld) + 8;

// REUSE (round 6): as in ilqr_lq_mfma.hip -- the env is time-invariant LQ and every pass runs at mu = 0, so K_t, V_xx(t), Q_uu(t) do not depend on the
// trajectory.  The first backward pass also leaves -Q_uu(t)^-1 in the workspace: lanes 49 .. 63 of the 16-pivot LDL^T solve (beyond the 32 + 16 + 1
// columns of [Q_ux | Q_uu | Q_u]: zero columns until now) read the identity columns e_0 .. e_14 and come out holding columns 0 .. 14 of the inverse;
// its last row is their sixteenth entries by symmetry, and its (15, 15) entry is -1 / d_15, the folded reciprocal of the last pivot (wave_ldlt.h).
// Every later pass runs the VECTOR recursion of ilqr.py:122-123,152-156 alone -- Q_x = l_x + F_x^T V_x, Q_u = l_u + F_u^T V_x, k = -Q_uu^-1 Q_u,
// V_x' = Q_x + K^T Q_u -- as wave-wide fp32 mat-vecs with K_t, Q_uu^-1, l_z(t) from four-deep register rings.  TFMPC_ILQR_LQ_REUSE=0: full pass always.
template <bool REUSE>
__global__ __launch_bounds__(kWave, 2) void ilqr_lq_mfma32_kernel(IlqrLqArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int i = lane & 15, q = lane >> 4;
    const int T = a.T, Tp = T + 1;
    const int n = a.env.n, m = a.env.m, d = n + m;
    const TfmpcIlqrConfig &cfg = a.cfg;
    const float *Fg = a.env.p[0] + (size_t)b * a.env.stride[0];
    const float *fg = a.env.p[1] + (size_t)b * a.env.stride[1];
    const float *Cg = a.env.p[2] + (size_t)b * a.env.stride[2];
    const float *cg = a.env.p[3] + (size_t)b * a.env.stride[3];
    float *Kg = a.wsK + (size_t)b * T * m * n;
    float *kg = a.wsk + (size_t)b * T * m;
    float *Mg = REUSE ? a.wsMinv + (size_t)b * T * (M * M) : nullptr;      // -Q_uu(t)^-1, [T][16][16] (padded actions: -1 on the diagonal)
    // trajectory buffers: [0] the output arrays, [1] the workspace; the nominal one is [flip]
    float *const xb[2] = {a.states + (size_t)b * Tp * n, a.wsx + (size_t)b * Tp * n};
    float *const ub[2] = {a.actions + (size_t)b * T * m, a.wsu + (size_t)b * T * m};
    float *const cb[2] = {a.costs + (size_t)b * Tp, a.wsc + (size_t)b * Tp};
    int flip = 0;

    auto zmap = [&](int zi) { return zi < N ? (zi < n ? zi : -1) : (zi - N < m ? n + zi - N : -1); };
    auto Fz = [&](int row, int zi) { const int c_ = zmap(zi); return (row < n && c_ >= 0) ? Fg[row * d + c_] : 0.0f; };
    auto Cs = [&](int zr, int zc) {            // symmetric part of C (gradient / Hessian of the cost), unit diagonal on padded actions
        const int r = zmap(zr), c_ = zmap(zc);
        if (r >= 0 && c_ >= 0) return 0.5f * (Cg[r * d + c_] + Cg[c_ * d + r]);
        return (zr == zc && zr >= N + m && zr < D) ? 1.0f : 0.0f;
    };
    auto cz = [&](int zr) { const int r = zmap(zr); return r >= 0 ? cg[r] : 0.0f; };

    // ---- operands of the sweep: bf16x3 fragments of F~ = [F_x | F_u] (no f column: the affine slot carries V_x) and the tiles of C_s.
    // Resident for the whole solve -- or, with REUSE, loaded inside the one pass that uses them (84 registers that are dead afterwards)
    ConstFrag Fc[2][3];
    f32x4 Cxx[2][2], Cux[2], Cuu;
    auto load_sweep_operands = [&]() {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = Fz(16 * kt + 4 * q + r, 16 * ct + i);
                Fc[kt][ct] = const_frag(v);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 2; ++b_) Cxx[a_][b_][r] = Cs(16 * a_ + 4 * q + r, 16 * b_ + i);
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) Cux[b_][r] = Cs(N + 4 * q + r, 16 * b_ + i);
            Cuu[r] = Cs(N + 4 * q + r, N + i);
        }
    };
    if (!REUSE) load_sweep_operands();
    const int kv_src = (i == 0) ? kkv + q : kZeros + q;

    // ---- C Z on the f32 matrix cores over the trajectory (xs, us), 16 timesteps per tile; row T carries u = 0.
    //   GRAD: l_z(t) = C_s z_t + c -> (gx[t][n], gu[t][m]);  else cost(t) = 1/2 z^T C z + c^T z -> out[t]; returns sum of costs
    // The operands (C as 3 x 12 A fragments, c) are re-read from L2 at every call: kept across the sweep they would cost
    // 48 registers of a file the sweep fills.
    auto cz_pass = [&](const float *xs, const float *us, bool grad, float *gx, float *gu, float *out) -> float {
        const int io = opaque(i), qo = opaque(q);
        float Ca[3][12];
        f32x4 cq[3];
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
#pragma unroll
            for (int s2 = 0; s2 < 12; ++s2) {
                const int r = zmap(16 * rt + io), c_ = zmap(4 * s2 + qo);
                Ca[rt][s2] = (r >= 0 && c_ >= 0) ? 0.5f * (Cg[r * d + c_] + Cg[c_ * d + r]) : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) cq[rt][r] = cz(16 * rt + 4 * qo + r);
        }
        float jsum = 0.0f;
        for (int nt = 0; nt * 16 < Tp; ++nt) {
            const int t = (16 * nt + io < Tp) ? 16 * nt + io : T;
            f32x4 Dz[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            float zv[12];
#pragma unroll
            for (int s2 = 0; s2 < 12; ++s2) {
                const int k = 4 * s2 + qo;
                zv[s2] = k < N ? (k < n ? xs[(size_t)t * n + k] : 0.0f) : ((t < T && k - N < m) ? us[(size_t)t * m + k - N] : 0.0f);
            }
#pragma unroll
            for (int s2 = 0; s2 < 12; ++s2)
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) Dz[rt] = mfma(Ca[rt][s2], zv[s2], Dz[rt]);
            const bool valid = 16 * nt + io < Tp;
            if (grad) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int r0 = 4 * qo + r;
                    if (valid && r0 < n) gx[(size_t)t * n + r0] = Dz[0][r] + cq[0][r];
                    if (valid && 16 + r0 < n) gx[(size_t)t * n + 16 + r0] = Dz[1][r] + cq[1][r];
                    if (valid && t < T && r0 < m) gu[(size_t)t * m + r0] = Dz[2][r] + cq[2][r];
                }
            } else {
                // z entries of this lane's rows (D layout rows 16 rt + 4 q + r of column t)
                float part = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int r0 = 4 * qo + r;
                    const float z0 = r0 < n ? xs[(size_t)t * n + r0] : 0.0f;
                    const float z1 = 16 + r0 < n ? xs[(size_t)t * n + 16 + r0] : 0.0f;
                    const float z2 = (t < T && r0 < m) ? us[(size_t)t * m + r0] : 0.0f;
                    part = fmaf(z0, fmaf(0.5f, Dz[0][r], cq[0][r]), part);
                    part = fmaf(z1, fmaf(0.5f, Dz[1][r], cq[1][r]), part);
                    part = fmaf(z2, fmaf(0.5f, Dz[2][r], cq[2][r]), part);
                }
                part += __shfl_xor(part, 16, kWave);
                part += __shfl_xor(part, 32, kWave);
                if (qo == 0 && valid) { out[t] = part; jsum += part; }
            }
        }
        return grad ? 0.0f : wave_sum(jsum);
    };

    // ---- one rollout from x0 into (xs, us): SEARCH: u_t = u_hat_t + alpha k_t + K_t (x_t - x_hat_t) (ilqr.py:193-197,
    // unbounded: no clip); else the injected start actions (:53-82).  LDS chunks of kTC timesteps, bulk stores.
    auto rollout = [&](bool search, float alpha, const float *xh, const float *uh, float *xs, float *us, float &residual) {
        const int lo = opaque(lane);
        const int fi = lo >> 1, fc = lo & 1;           // F: row fi, z columns 24 fc .. 24 fc + 23
        const int ka = lo >> 2, jc = lo & 3;           // K: row ka, columns 8 jc .. 8 jc + 7
        float Fr[24];
#pragma unroll
        for (int j = 0; j < 24; ++j) Fr[j] = Fz(fi, 24 * fc + j);
        const float f_part = (fc == 0 && fi < n) ? fg[fi] : 0.0f;
        float *zs = &lds[0];
        __syncthreads();
        if (lane < N) {
            const float x = lane < n ? a.x0[(size_t)b * n + lane] : 0.0f;
            zs[lane] = x;
            if (lane < n) xs[lane] = x;
        }
        // inputs of step t for this lane: K[ka][8 jc ..], k[ka], x_hat[8 jc ..], u_hat[ka]
        auto load_step = [&](int t, float (&Kv)[8], float (&xv)[8], float &kv, float &uv) {
            if (n == N && m == M && ((reinterpret_cast<uintptr_t>(xh) | reinterpret_cast<uintptr_t>(Kg)) & 15u) == 0) {
                // the literal shape (BASELINE configs[4]), 16-byte aligned trajectories: no bounds to test, 16-byte loads
                uv = uh[(size_t)t * M + ka];
                if (search) {
                    const f32x4 *Kp = reinterpret_cast<const f32x4 *>(Kg + (size_t)t * (M * N) + ka * N + 8 * jc);
                    const f32x4 *xp = reinterpret_cast<const f32x4 *>(xh + (size_t)t * N + 8 * jc);
                    const f32x4 K0 = Kp[0], K1 = Kp[1], x0v = xp[0], x1v = xp[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { Kv[j] = K0[j]; Kv[4 + j] = K1[j]; xv[j] = x0v[j]; xv[4 + j] = x1v[j]; }
                    kv = kg[(size_t)t * M + ka];
                }
                return;
            }
            const bool row = ka < m;
            uv = row ? uh[(size_t)t * m + ka] : 0.0f;
            if (search) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool col = 8 * jc + j < n;
                    Kv[j] = (row && col) ? Kg[(size_t)t * m * n + ka * n + 8 * jc + j] : 0.0f;
                    xv[j] = col ? xh[(size_t)t * n + 8 * jc + j] : 0.0f;
                }
                kv = row ? kg[(size_t)t * m + ka] : 0.0f;
            }
        };
        // inputs of the coming steps in a register ring, statically indexed through the unrolled inner loop and refilled unconditionally (wave_ops.h,
        // "two rules of the time loops"): four steps deep where the registers are there (REUSE: the sweep's operands are dead by now), else one
        constexpr int kRing = REUSE ? 4 : 1;
        static_assert(kTC % kRing == 0, "a chunk holds whole turns of the ring");
        float KR[kRing][8], xR[kRing][8], kR[kRing], uR[kRing];
#pragma unroll
        for (int d = 0; d < kRing; ++d) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { KR[d][j] = 0.0f; xR[d][j] = 0.0f; }
            kR[d] = 0.0f; uR[d] = 0.0f;
            if (T > 0) load_step(d < T ? d : T - 1, KR[d], xR[d], kR[d], uR[d]);
        }
        float rmax = 0.0f;
        __syncthreads();
        for (int t0 = 0; t0 < T; t0 += kTC) {
            const int tc = (T - t0 < kTC) ? (T - t0) : kTC;
            for (int tb = 0; tb < tc; tb += kRing) {
#pragma unroll
                for (int d = 0; d < kRing; ++d) {
                    const int tt = tb + d;
                    if (tt >= tc) break;
                    const int t = t0 + tt;
                    float *zt = zs + tt * kZld;
                    float Kc[8], xc[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { Kc[j] = KR[d][j]; xc[j] = xR[d][j]; }
                    const float kc = kR[d], uc = uR[d];
                    load_step(t + kRing < T ? t + kRing : T - 1, KR[d], xR[d], kR[d], uR[d]);
                    float u = uc;
                    if (search) {
                        const f32x4 xlo = *reinterpret_cast<const f32x4 *>(&zt[8 * jc]), xhi = *reinterpret_cast<const f32x4 *>(&zt[8 * jc + 4]);
                        float du = 0.0f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { du = fmaf(Kc[j], xlo[j] - xc[j], du); du = fmaf(Kc[4 + j], xhi[j] - xc[4 + j], du); }   // K (x - x_hat)
                        du += dpp<kDppXor1>(du);
                        du += dpp<kDppXor2>(du);
                        du = fmaf(alpha, kc, du);
                        rmax = fmaxf(rmax, fabsf(du));                                     // :206
                        u = uc + du;
                    }
                    zt[N + ka] = u;                      // the four lanes of the row hold the same value
                    lds_sync();
                    float xn = f_part;                   // x' = F z + f
#pragma unroll
                    for (int j4 = 0; j4 < 6; ++j4) {
                        const f32x4 z4 = *reinterpret_cast<const f32x4 *>(&zt[24 * fc + 4 * j4]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) xn = fmaf(Fr[4 * j4 + j], z4[j], xn);
                    }
                    xn += dpp<kDppXor1>(xn);
                    zt[kZld + fi] = xn;
                    lds_sync();
                }
            }
            for (int idx = lane; idx < tc * n; idx += kWave) xs[(size_t)(t0 + 1) * n + idx] = zs[(1 + idx / n) * kZld + idx % n];
            for (int idx = lane; idx < tc * m; idx += kWave) us[(size_t)t0 * m + idx] = zs[(idx / m) * kZld + N + idx % m];
            __syncthreads();
            if (lane < N) zs[lane] = zs[tc * kZld + lane];      // carry x into row 0 of the next chunk
            __syncthreads();
        }
        residual = wave_max(rmax);
        __syncthreads();                               // the trajectory stores are complete before cz_pass reads them back
    };

    // ---- start (ilqr.py:218) ------------------------------------------------------------------------------------
    float dummy;
    rollout(false, 0.0f, nullptr, a.u_init + (size_t)b * T * m, xb[0], ub[0], dummy);
    float J_hat = cz_pass(xb[0], ub[0], false, nullptr, nullptr, cb[0]);

    int status = 0, iteration = 0;
    bool converged = false, retry = false;
    float delta = 1.0f;                                                // :216 (mu stays 0 in this kernel; delta is only logged)
    // One iteration = derivatives, backward pass, line search (ilqr.py:234-279) as pieces, so that with REUSE the first iteration (full pass) stands
    // outside the loop of the later ones (vector recursion) and the sweep's 84 operand registers are dead while those run.
    float dV1 = 0.0f, gsum = 0.0f;
    float *xhat = nullptr, *uhat = nullptr, *xc = nullptr, *uc = nullptr, *cc = nullptr, *Lx = nullptr, *Lu = nullptr;
    auto derivatives = [&]() {
        xhat = xb[flip]; uhat = ub[flip];
        xc = xb[flip ^ 1]; uc = ub[flip ^ 1]; cc = cb[flip ^ 1];
        Lx = xc; Lu = uc;
        // ---- derivatives (ilqr.py:234): l_z(t) of the nominal trajectory, parked in the candidate buffers ---------------
        cz_pass(xhat, uhat, true, Lx, Lu, nullptr);
        for (int idx = lane; idx < kSweepFloats; idx += kWave) lds[idx] = 0.0f;
        if (REUSE && lane < M - 1) lds[kMs + lane * kMld + N + M + 1 + lane] = 1.0f;      // e_0 .. e_14 in the columns of lanes 49 .. 63 (same wave: after the zeroing)
        __syncthreads();

    };
    auto sweep_pass = [&]() -> bool {            // false: Q_uu not positive definite (needs mu > 0: the wave kernel re-solves the instance)
        if (REUSE) load_sweep_operands();
        // ---- backward (ilqr.py:94-172 with mu = 0): the sweep of lqr_mfma32x16.hip ----------------------------------------
        f32x4 Vd[2][2], vd[2];
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) Vd[a_][b_] = Cxx[a_][b_];
        // column loads of the affine slot: lanes i == 0 rows 16 a + 4 q + r
        auto load_col = [&](const float *p, int count, int base) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (base + 4 * q + r < count) v[r] = p[base + 4 * q + r];
            }
            return v;
        };
        vd[0] = load_col(Lx + (size_t)T * n, n, 0);                            // V_x = l_x^f
        vd[1] = load_col(Lx + (size_t)T * n, n, 16);
        int min_pivot_bits = 0x3f800000;
        dV1 = 0.0f; gsum = 0.0f;
        f32x4 lxn[2], lun;                                                       // l_x(t), l_u(t) one step ahead
        if (T > 0) {
            lxn[0] = load_col(Lx + (size_t)(T - 1) * n, n, 0);
            lxn[1] = load_col(Lx + (size_t)(T - 1) * n, n, 16);
            lun = load_col(Lu + (size_t)(T - 1) * m, m, 0);
        }
        for (int t = T - 1; t >= 0; --t) {
            const f32x4 lx0 = lxn[0], lx1 = lxn[1], lu = lun;
            if (t > 0) {
                lxn[0] = load_col(Lx + (size_t)(t - 1) * n, n, 0);
                lxn[1] = load_col(Lx + (size_t)(t - 1) * n, n, 16);
                lun = load_col(Lu + (size_t)(t - 1) * m, m, 0);
            }
            // 1. W = V_xx [F_x | F_u]; the affine tile is V_x itself
            f32x4 W[2][3];
            {
                VarFrag Vf[2][2];
#pragma unroll
                for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                    for (int b_ = 0; b_ < 2; ++b_) Vf[a_][b_] = var_frag(Vd[a_][b_]);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) acc = mm_var_const(Vf[kt][rt], Fc[kt][ct], acc);
                        W[rt][ct] = acc;
                    }
            }
            // 2. Q terms (:122-131)
            f32x4 Qxx[2][2], Qux[2], qx[2], Quu, qu;
            {
                VarFrag Wf[2][3], vf[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) Wf[kt][ct] = var_frag(W[kt][ct]);
                    vf[kt] = var_frag(vd[kt]);
                }
#pragma unroll
                for (int a_ = 0; a_ < 2; ++a_) {
#pragma unroll
                    for (int b_ = 0; b_ < 2; ++b_) {
                        f32x4 acc = Cxx[a_][b_];
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) acc = mm_const_var(Fc[kt][a_], Wf[kt][b_], acc);
                        Qxx[a_][b_] = acc;                                         // Q_xx                 :129
                    }
                    f32x4 acc = a_ == 0 ? lx0 : lx1;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) acc = mm_const_var(Fc[kt][a_], vf[kt], acc);
                    qx[a_] = acc;                                                  // Q_x = l_x + F_x^T V_x  :122
                }
#pragma unroll
                for (int b_ = 0; b_ < 2; ++b_) {
                    f32x4 acc = Cux[b_];
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) acc = mm_var_const(Wf[kt][2], Fc[kt][b_], acc);
                    Qux[b_] = acc;                                                 // Q_ux                 :131
                }
                f32x4 acc = Cuu;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) acc = mm_const_var(Fc[kt][2], Wf[kt][2], acc);
                Quu = acc;                                                         // Q_uu                 :130
                acc = lu;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) acc = mm_const_var(Fc[kt][2], vf[kt], acc);
                qu = acc;                                                          // Q_u = l_u + F_u^T V_x  :123
            }
            // 3. elimination input, row-major staging -> one column per lane
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lds[kMs + (4 * q + r) * kMld + i] = Qux[0][r];
                lds[kMs + (4 * q + r) * kMld + 16 + i] = Qux[1][r];
                lds[kMs + (4 * q + r) * kMld + N + i] = Quu[r];
                if (i == 0) lds[kMs + (4 * q + r) * kMld + N + M] = qu[r];
            }
            __syncthreads();
            f32x2 M2[M / 2];
            float quc[M];                                                           // column of this lane before the elimination (lane 48: Q_u)
#pragma unroll
            for (int e = 0; e < M / 2; ++e) {
                M2[e] = f32x2{lds[kMs + (2 * e) * kMld + lane], lds[kMs + (2 * e + 1) * kMld + lane]};
                quc[2 * e] = M2[e][0];
                quc[2 * e + 1] = M2[e][1];
            }
            float Mr[M];
            float ninv_last = 0.0f;
            ldlt_solve_neg<M, N>(M2, Mr, min_pivot_bits, REUSE ? &ninv_last : nullptr);   // [K | k] = -Q_uu^-1 [Q_ux | Q_u]  :357-362
            if (REUSE && lane > N + M) {                  // lanes 49 .. 63: column j = lane - 49 of -Q_uu^-1 = its row j; their 16th entries = row 15
                float *Mt = Mg + (size_t)t * (M * M);
                const int j = lane - (N + M + 1);
#pragma unroll
                for (int e = 0; e < M; e += 4) *reinterpret_cast<f32x4 *>(&Mt[j * M + e]) = f32x4{Mr[e], Mr[e + 1], Mr[e + 2], Mr[e + 3]};
                Mt[(M - 1) * M + j] = Mr[M - 1];
                if (lane == 63) Mt[M * M - 1] = ninv_last;
            }
            {   // dV1 += k^T Q_u (:166), g_norm term max_a |k_a| / (|u_hat_a| + 1) (:243): lane 48 holds k and Q_u
                float p1 = 0.0f, gm = 0.0f;
#pragma unroll
                for (int e = 0; e < M; ++e) {
                    p1 = fmaf(Mr[e], quc[e], p1);
                    const float uh = e < m ? uhat[(size_t)t * m + e] : 0.0f;
                    // (hardware reciprocal, 1 ulp, instead of an IEEE division -- sixteen per step were ~130 instructions: g_norm is
                    // only ever compared with atol; the costate kernels do the same)
                    gm = fmaxf(gm, e < m ? fabsf(Mr[e]) * __builtin_amdgcn_rcpf(fabsf(uh) + 1.0f) : 0.0f);
                }
                dV1 += readlane(p1, N + M);
                gsum += readlane(gm, N + M);
            }
            if (lane < N) {
#pragma unroll
                for (int e = 0; e < M; ++e) lds[kKs + e * N + lane] = Mr[e];
            } else if (lane == N + M) {
#pragma unroll
                for (int e = 0; e < M; e += 4) *reinterpret_cast<f32x4 *>(&lds[kkv + e]) = f32x4{Mr[e], Mr[e + 1], Mr[e + 2], Mr[e + 3]};
            }
            __syncthreads();
            // 4. V_xx' = Q_xx + Q_xu K, V_x' = Q_x + Q_xu k                              :149-161
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const float ax0 = lds[kMs + (4 * s2 + q) * kMld + i];
                const float ax1 = lds[kMs + (4 * s2 + q) * kMld + 16 + i];
                const float g0 = lds[kKs + (4 * s2 + q) * N + i];
                const float g1 = lds[kKs + (4 * s2 + q) * N + 16 + i];
                const float gk = lds[kv_src + 4 * s2];
                Qxx[0][0] = mfma(ax0, g0, Qxx[0][0]);
                Qxx[0][1] = mfma(ax0, g1, Qxx[0][1]);
                Qxx[1][0] = mfma(ax1, g0, Qxx[1][0]);
                Qxx[1][1] = mfma(ax1, g1, Qxx[1][1]);
                qx[0] = mfma(ax0, gk, qx[0]);
                qx[1] = mfma(ax1, gk, qx[1]);
            }
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
                    for (int r = 0; r < 4; ++r) lds[kVt + (16 * a_ + 4 * q + r) * kVld + 16 * b_ + i] = Qxx[a_][b_][r];
            __syncthreads();
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
                    for (int r = 0; r < 4; ++r)                                   // V_xx <- (V_xx + V_xx^T) / 2   :158-162
                        Vd[a_][b_][r] = 0.5f * (Qxx[a_][b_][r] + lds[kVt + (16 * b_ + i) * kVld + 16 * a_ + 4 * q + r]);
            vd[0] = qx[0];
            vd[1] = qx[1];
            // gains to HBM, row-major K[t][a][j], k[t][a]
            if (n == N && m == M) {
                float *Kt = Kg + (size_t)t * (M * N);
#pragma unroll
                for (int w = 0; w < 2; ++w) *reinterpret_cast<f32x4 *>(&Kt[4 * (lane + 64 * w)]) = *reinterpret_cast<const f32x4 *>(&lds[kKs + 4 * (lane + 64 * w)]);
                if (lane < M) kg[(size_t)t * M + lane] = lds[kkv + lane];
            } else {
                for (int idx = lane; idx < M * N; idx += kWave) {
                    const int ka = idx >> 5, j = idx & 31;
                    if (ka < m && j < n) Kg[(size_t)t * m * n + ka * n + j] = lds[kKs + idx];
                }
                if (lane < m) kg[(size_t)t * m + lane] = lds[kkv + lane];
            }
            __syncthreads();
        }
        if (min_pivot_bits <= 0) { status |= TFMPC_ST_NOT_PD; retry = true; return false; }   // needs mu > 0
        return true;
    };
    auto vector_pass = [&]() {
        // ---- the vector recursion alone (REUSE, iterations >= 1): K_t and -Q_uu(t)^-1 of the first pass are this pass's too -----------------------
        constexpr int kVx = kMs, kQx = kMs + 32, kQu = kMs + 64;      // V_x(32) | Q_x(32) | Q_u(16) staging (the elimination buffers are idle)
        const int lo = opaque(lane);
        const int o = lo >> 2, part = lo & 3;              // layout A: 16 outputs x 4 contraction parts (F~^T V_x in three rounds; Q_uu^-1 Q_u)
        const int o2 = lo >> 1, half = lo & 1;             // layout B: 32 outputs x 2 halves (K^T Q_u)
        float Ft[3][8];                                    // F~[8 part + j][16 r + o]
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) Ft[r][j] = Fz(8 * part + j, 16 * r + o);
        const bool exact = n == N && m == M && ((reinterpret_cast<uintptr_t>(Lx) | reinterpret_cast<uintptr_t>(Kg)) & 15u) == 0;
        // per-step inputs from HBM: l_z(t) in layout A (three per lane), four entries of row o of -Q_uu^-1, eight of column o2 of K_t
        auto load_step = [&](int t, float (&lz)[3], f32x4 &Mi, float (&Kt)[8]) {
            if (exact) {
                const float *Lxt = Lx + (size_t)t * N, *Lut = Lu + (size_t)t * M, *Kr = Kg + (size_t)t * (M * N);
                lz[0] = Lxt[o]; lz[1] = Lxt[16 + o]; lz[2] = Lut[o];
#pragma unroll
                for (int j = 0; j < 8; ++j) Kt[j] = Kr[(8 * half + j) * N + o2];
            } else {
                lz[0] = o < n ? Lx[(size_t)t * n + o] : 0.0f;
                lz[1] = 16 + o < n ? Lx[(size_t)t * n + 16 + o] : 0.0f;
                lz[2] = o < m ? Lu[(size_t)t * m + o] : 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) Kt[j] = (8 * half + j < m && o2 < n) ? Kg[(size_t)t * m * n + (8 * half + j) * n + o2] : 0.0f;
            }
            Mi = *reinterpret_cast<const f32x4 *>(Mg + (size_t)t * (M * M) + 4 * lo);
        };
        // V_x = l_x^f                                                                    :113
        if (lane < N) lds[kVx + lane] = lane < n ? Lx[(size_t)T * n + lane] : 0.0f;
        constexpr int kDepth = 4;
        float lzR[kDepth][3], KtR[kDepth][8];
        f32x4 MiR[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) load_step(T - 1 - d >= 0 ? T - 1 - d : 0, lzR[d], MiR[d], KtR[d]);
        float p1 = 0.0f;
        __syncthreads();
        for (int tb = T - 1; tb >= 0; tb -= kDepth) {
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {
                const int t = tb - d;
                if (t < 0) break;
                float lz[3], Kt[8];
#pragma unroll
                for (int j = 0; j < 3; ++j) lz[j] = lzR[d][j];
#pragma unroll
                for (int j = 0; j < 8; ++j) Kt[j] = KtR[d][j];
                const f32x4 Mi = MiR[d];
                load_step(t - kDepth >= 0 ? t - kDepth : 0, lzR[d], MiR[d], KtR[d]);       // unconditional: the compiler can count the loads in flight
                const f32x4 va = *reinterpret_cast<const f32x4 *>(&lds[kVx + 8 * part]), vb = *reinterpret_cast<const f32x4 *>(&lds[kVx + 8 * part + 4]);
                float y[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    float acc = Ft[r][0] * va[0];
#pragma unroll
                    for (int j = 1; j < 4; ++j) acc = fmaf(Ft[r][j], va[j], acc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = fmaf(Ft[r][4 + j], vb[j], acc);
                    acc += dpp<kDppXor1>(acc);
                    acc += dpp<kDppXor2>(acc);
                    y[r] = lz[r] + acc;                                                  // Q_x[o], Q_x[16 + o], Q_u[o]      :122-123
                }
                if (part == 0) { lds[kQx + o] = y[0]; lds[kQx + 16 + o] = y[1]; lds[kQu + o] = y[2]; }
                lds_sync();
                const f32x4 qa = *reinterpret_cast<const f32x4 *>(&lds[kQu + 4 * part]);
                float kk = Mi[0] * qa[0];                                                // k = -Q_uu^-1 Q_u                   :357-362
#pragma unroll
                for (int j = 1; j < 4; ++j) kk = fmaf(Mi[j], qa[j], kk);
                kk += dpp<kDppXor1>(kk);
                kk += dpp<kDppXor2>(kk);
                const f32x4 q0 = *reinterpret_cast<const f32x4 *>(&lds[kQu + 8 * half]), q1 = *reinterpret_cast<const f32x4 *>(&lds[kQu + 8 * half + 4]);
                float w = Kt[0] * q0[0];
#pragma unroll
                for (int j = 1; j < 4; ++j) w = fmaf(Kt[j], q0[j], w);
#pragma unroll
                for (int j = 0; j < 4; ++j) w = fmaf(Kt[4 + j], q1[j], w);
                w += dpp<kDppXor1>(w);                                                   // (K^T Q_u)[o2] = (Q_xu k)[o2]
                const float vnew = lds[kQx + o2] + w;                                    // V_x' = Q_x + Q_xu k                :152-156
                if (part == 0) {
                    p1 = fmaf(kk, y[2], p1);                                             // dV1 += k^T Q_u                     :166
                    if (o < m) kg[(size_t)t * m + o] = kk;
                }
                lds_sync();                                                              // (every lane has read Q_x before V_x is replaced)
                if (half == 0) lds[kVx + o2] = vnew;
                lds_sync();
            }
        }
        __syncthreads();                                   // k_t (global memory) is read by other lanes below
        dV1 = wave_sum(p1);
        // g_norm term: mean_t max_a |k_a| / (|u_hat_a| + 1) (:243), sixteen lanes per time step (hardware reciprocal as in the full pass)
        float gs = 0.0f;
        for (int base = 0; base < T * M; base += kWave) {
            const int idx = base + lane, t = idx >> 4, ua = idx & 15;
            float ratio = 0.0f;
            if (idx < T * M && ua < m) ratio = fabsf(kg[(size_t)t * m + ua]) * __builtin_amdgcn_rcpf(fabsf(uhat[(size_t)t * m + ua]) + 1.0f);
            ratio = fmaxf(ratio, dpp<kDppXor1>(ratio));
            ratio = fmaxf(ratio, dpp<kDppXor2>(ratio));
            ratio = fmaxf(ratio, dpp<0x141>(ratio));                                     // row_half_mirror
            ratio = fmaxf(ratio, dpp<0x140>(ratio));                                     // row_mirror: every lane of the row holds the row's maximum
            if (ua == 0 && idx < T * M) gs += ratio;
        }
        gsum = wave_sum(gs);
    };
    auto finish_pass = [&]() -> bool {           // true: the solve of this instance ends here (converged, or handed to the wave kernel)
        const float dV2 = -0.5f * dV1;                                                 // :167 at mu = 0
        const float g_norm = T > 0 ? gsum / (float)T : 0.0f;
        // decision trace (what ilqr.py:243-279 logs per pass): mu = 0 in every pass of this kernel, so row = iteration; an
        // instance handed to the wave kernel (retry) has its rows rewritten by that kernel
        if (g_norm < cfg.atol) {                                                       // :243-248
            if (lane == 0) trace_write(a.trace, b, iteration, iteration, 0.0f, delta, J_hat, g_norm, -1, 0.0f, 0.0f, -1, -1.0f);
            converged = true;
            return true;
        }

        // ---- forward / line search (ilqr.py:317-355) -------------------------------------------------------------------
        bool accept = false;
        float residual = 0.0f, J = 0.0f;
        int ai_last = -1;
        for (int ai = 0; ai < cfg.n_alphas; ++ai) {
            const float alpha = cfg.alphas[ai];
            ai_last = ai;
            rollout(true, alpha, xhat, uhat, xc, uc, residual);
            J = cz_pass(xc, uc, false, nullptr, nullptr, cc);
            const float delta_J = -alpha * (dV1 + alpha * dV2);                    // :339
            const float dcost = J_hat - J;
            const float z = (delta_J > 0.0f) ? dcost / delta_J : sgn(dcost);       // :342-346
            if (z >= cfg.c1) { accept = true; break; }                             // :351-353
        }
        const bool small_step = residual < cfg.atol;                              // :253-257
        if (lane == 0)
            trace_write(a.trace, b, iteration, iteration, 0.0f, delta, J_hat, g_norm, ai_last,
                        ai_last >= 0 ? cfg.alphas[ai_last] : 0.0f, J, accept ? 1 : 0, residual);
        if (small_step || accept) { flip ^= 1; J_hat = J; }                        // the candidate becomes the nominal
        if (small_step) { converged = true; return true; }
        if (!accept) { retry = true; return true; }                                      // would raise mu (:267-270)
        delta = fminf(1.0f / cfg.delta_0, delta / cfg.delta_0);                    // :259-266 at mu = 0
        return false;
    };
    if (REUSE) {
        derivatives();
        bool stop = !sweep_pass();
        if (!stop) stop = finish_pass();
        if (!stop) {
            for (iteration = 1; iteration < cfg.max_iterations; ++iteration) {
                derivatives();
                vector_pass();
                if (finish_pass()) break;
            }
        }
    } else {
        for (iteration = 0; iteration < cfg.max_iterations; ++iteration) {
            derivatives();
            if (!sweep_pass()) break;
            if (finish_pass()) break;
        }
    }
    if (iteration >= cfg.max_iterations) iteration = cfg.max_iterations - 1;
    (void)converged;

    // ---- results: a nominal trajectory that ended in the workspace is copied out ----------------------------------------
    __syncthreads();
    if (flip) {
        for (int idx = lane; idx < Tp * n; idx += kWave) xb[0][idx] = xb[1][idx];
        for (int idx = lane; idx < T * m; idx += kWave) ub[0][idx] = ub[1][idx];
        for (int idx = lane; idx < Tp; idx += kWave) cb[0][idx] = cb[1][idx];
    }
    if (lane == 0) {
        if (!(J_hat == J_hat)) status |= TFMPC_ST_NAN;
        if (retry) status |= kIlqrRetryBit;
        a.iterations[b] = iteration;
        a.status[b] = status;
    }
}

}  // namespace

bool ilqr_lq_mfma32_supported(const TfmpcEnv &env, int T)
{
    return env.kind == TFMPC_ENV_LQ && !env.bounded && !env.any_finite_bound && env.n <= N && env.m <= M &&
           !(env.n <= 16 && env.m <= 8) && env.n + env.m >= 12 && T >= 1;
}

size_t ilqr_lq_mfma32_reuse_workspace_bytes(int B, int n, int m, int T)
{
    if (!(n <= N && m <= M && !(n <= 16 && m <= 8) && n + m >= 12) || T < 1) return 0;      // the shapes ilqr_lq_mfma32_supported admits
    return (size_t)B * T * (M * M) * sizeof(float);
}

int ilqr_lq_mfma32_launch(const IlqrLqArgs &a, hipStream_t stream)
{
    const bool reuse = a.wsMinv != nullptr && (reinterpret_cast<uintptr_t>(a.wsMinv) & 15u) == 0 && !option_is(kOptIlqrLqReuse, "0");
    if (reuse) hipLaunchKernelGGL(ilqr_lq_mfma32_kernel<true>, dim3(a.B), dim3(kWave), 0, stream, a);
    else hipLaunchKernelGGL(ilqr_lq_mfma32_kernel<false>, dim3(a.B), dim3(kWave), 0, stream, a);
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}

}  // namespace tfmpc
