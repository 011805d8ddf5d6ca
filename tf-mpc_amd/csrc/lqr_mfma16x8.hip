// lqr_mfma16x8.hip -- LQR backward + forward for the BASELINE.json headline shape
// (state_dim n = 16, action_dim m = 8, any horizon) on gfx950 matrix cores; smaller shapes
// (n <= 16, m <= 8) run through the same kernel zero-padded to 16 x 8 (EXACT = false).
//
// Replaces tfmpc/solvers/lqr.py:59-166 of the reference for that shape.  One
// wavefront owns one problem instance for the whole solve.
//
// Backward sweep (lqr.py:73-127), per timestep, all in registers:
//   F~ = [F_x | F_u f 0] (16 x 32) and the tiles of C~ = [[C, c],[.,0]] are loaded once in
//   matrix-core operand layout and stay resident for all T steps.
//   1. W = V F~ (two 16 x 16 tiles; column 24 of W is V f; += v).
//   2. THREE 16 x 16 tiles of Q~ = C~ + F~^T W:  Q_xx = F_x^T W_0;  [Q_uu | q_u] = F~_1^T W_1;
//      and W_1^T F_x, whose rows 0..7 are Q_ux and whose row 8 is q_x^T (V is kept exactly
//      symmetric, step 4, so the F~_1^T W_0 tile is redundant).
//      The accumulator layout of a 16x16 MFMA (lane (j, q) holds rows 4q..4q+3 of column j) IS the
//      operand layout of the next product when the contraction index is enumerated k = 4q + r:
//      W feeds step 2 and V feeds step 1 with no data movement, and one register set of F~ is
//      the B operand of step 1 and the A operand (F~^T) of step 2.
//      Default (BF3): each of these five products is evaluated as "bf16x3" on the bf16 matrix
//      cores -- every fp32 operand split into three bf16 parts (24 mantissa bits), six partial
//      products carried by three v_mfma_f32_16x16x32_bf16 with fp32 accumulation
//      (mfma_bf16x3.h): 15 MFMAs per step.  TFMPC_LQR_MFMA=f32 (tfmpc_set_option) keeps them on
//      v_mfma_f32_16x16x4_f32: 20 MFMAs per step, exact fp32 FMA chains.
//   3. [Q_ux | Q_uu | q_u] (8 x 25) crosses LDS once into "one column per lane, eight rows in
//      registers" and is solved by an LDL^T elimination WITHOUT pivoting that reads only the
//      upper triangle of Q_uu (wave_ldlt8.h; one v_readlane per multiplier serves the forward and
//      the backward sweep) -> K~ = -Q_uu^-1 [Q_ux | q_u].  The reference takes a general
//      inverse (lqr.py:84-87); for a symmetric C with C_uu > 0 (the precondition stated in
//      include/tfmpc_hip.h) Q_uu is SPD and the two agree to rounding.  A non-positive pivot is
//      reported in status[b] (TFMPC_ST_NOT_PD / _SINGULAR).
//   4. V' = Q_xx + Q_xu K, v' = q_x + Q_xu k: 4 x v_mfma_f32_16x16x4_f32 (the Schur-complement
//      form of the four-term update lqr.py:97-105, equal to it in exact arithmetic), then
//      V' <- (V' + V'^T) / 2 through an LDS transpose: steps 2 and 3 use the symmetry of V, so its
//      rounding-level antisymmetric part must not survive a step (it would grow like |F_u K|^2
//      per step; the reference does not symmetrise, quirk Q6 of SURVEY.md).
//   K_t, k_t stream to HBM (row-major, the public K/k layout) for the rollout.
// Forward rollout (lqr.py:141-155): wave-wide fp32 FMA mat-vecs with F rows resident in
//   registers, z_t = [x_t; u_t] staged in LDS, K_t prefetched one step ahead, DPP reductions;
//   the stage costs are priced afterwards as C Z on the matrix cores, states / actions leave in
//   bulk coalesced stores.
//
// The f32 MFMA is an exact fp32 FMA chain (MI355X_MICROARCH.md) and bf16x3 drops only terms below
// 2^-24 relative, so numerics are those of fp32 VALU code with a different summation order.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "lqr_kernels.h"
#include "options.h"
#include "mfma_bf16x3.h"
#include "wave_ldlt8.h"
#include "wave_ops.h"

namespace tfmpc {

namespace {

constexpr int N = 16, M = 8, D = 24;
using f32x4 = bf3::f32x4;

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float readlane(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

using namespace bf3;      // bf16x3 product helpers (mfma_bf16x3.h)

// per-wave LDS slice (floats)
constexpr int kMs = 0;          // [32 cols][8 rows]  elimination input, column-major
constexpr int kKs = 256;        // [32 cols][8 rows]  K~ = -Q_uu^-1 [Q_ux | . | q_u], column-major
constexpr int kZs = 512;        // rollout chunk: rows z_t = [x_t(16); u_t(8)], stride kZld (sweep: V' transpose staging)
constexpr int kZld = 26;        // even (8-byte aligned rows), 26 n mod 32 distinct for n < 16
constexpr int kVtLd = 20;        // V' transpose staging [16 cols][kVtLd] inside the rollout buffer
#ifndef TFMPC_LQR_TC
#define TFMPC_LQR_TC 52
#endif
constexpr int kTC = TFMPC_LQR_TC;         // timesteps per rollout chunk (T = 50 fits one chunk of 52)
constexpr int kLdsFloats = kZs + (kTC + 1) * kZld + 6;
#ifndef TFMPC_LQR_RING
#define TFMPC_LQR_RING 4
#endif
constexpr int kRing = TFMPC_LQR_RING;     // rollout: gains in flight, in steps (the chunk length is a multiple of it: slots keep their phase across chunks)
static_assert(kTC % kRing == 0, "a chunk must hold whole turns of the gain ring");

// DPP lane exchanges inside a row of 16 lanes (no LDS traffic, folded into the VALU op)
template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // lane i <-> 7 - i inside each 8-lane half row

// EXACT: n == 16 and m == 8 (no guards, vector gain stores).  Otherwise the instance is
// embedded in the 16 x 8 tile grid: states n..15 and actions m..7 are zero rows/columns of
// F~ and C~, with a unit diagonal on the padded part of C_uu so the elimination stays regular
// (the padded gains come out exactly 0).
// OUT16: the 16-bit copies of the policy / value outputs (LqrArgs::K16 ...) are compiled into separate instantiations, so
// that the default kernels carry none of their code
// Register budget (round 4).  Left to itself the compiler takes 105 - 109 VGPRs for the solve kernels: FOUR waves per SIMD (512 / 112).  Sized
// for FIVE (<= 96) the kernels without value-function outputs need 74 - 89 registers and spill nothing -- the LDS slice (7.6 KB) lets 21 waves
// into a CU -- and the headline launch gains 2 - 4 % on every box tried (tools/probes/r4_headline_eu.sh, alternating: 1.787 -> 1.744 ms, 1.86 ->
// 1.83 ms).  The VALUE instantiations would spill 2 - 19 registers at that budget and keep four.  -DTFMPC_LQR_EU=n forces a budget (A/B builds).
// Round 5: the budget is a template parameter (EU = resident waves per SIMD the register allocation is sized for) and TFMPC_LQR_WAVES=4|5
// picks the instantiation at run time, because round 4's verdict suspected the shard sizes of a strong-scaling run to prefer four: 8 192
// instances (the 8-GPU shard of the headline batch) are 8 waves per SIMD = 5 + 3 at five resident, 4 + 4 at four.  Measured on every shard
// size from 1 024 to 65 536 instances (tools/probes/r5_headline_shard_sweep.py, profiles/r05_headline_shard_sweep.json): FIVE wins at each
// of them (8 192: 0.238 against 0.258 ms; the kernel's time is 27.2 us per wave of a SIMD + ~20 us, the rounds overlap because waves do not
// finish together), so five stays the rule for every launch.  Same instruction stream per wave up to register allocation: bit-identical.
#ifdef TFMPC_LQR_EU
#define TFMPC_LQR_WAVES(EU_) TFMPC_LQR_EU
#else
#define TFMPC_LQR_WAVES(EU_) (EU_)
#endif
#define TFMPC_LQR_OCCUPANCY __attribute__((amdgpu_waves_per_eu(TFMPC_LQR_WAVES(EU), TFMPC_LQR_WAVES(EU))))
template <bool BACKWARD, bool FORWARD, bool VALUE, bool EXACT, bool BF3, bool OUT16 = false, int EU = 4>
__global__ __launch_bounds__(kWave) TFMPC_LQR_OCCUPANCY void lqr_mfma16x8_kernel(LqrArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int i = lane & 15, q = lane >> 4;
    const int T = a.T;
    const int n = EXACT ? N : a.n, m = EXACT ? M : a.m, d = n + m;
    // padded-index accessors: x index xi in [0,16), u index ui in [0,8)
    auto Fxx = [&](const float *Fg, int row, int xi) { return (row < n && xi < n) ? Fg[row * d + xi] : 0.0f; };
    auto Fxu = [&](const float *Fg, int row, int ui) { return (row < n && ui < m) ? Fg[row * d + n + ui] : 0.0f; };
    // z index zi in [0,24): 0..15 -> x, 16..23 -> u; returns the real row/col of C, c or -1
    auto zmap = [&](int zi) { return zi < N ? (zi < n ? zi : -1) : (zi - N < m ? n + zi - N : -1); };
    auto Czz = [&](const float *Cg, int zr, int zc) {
        const int r = zmap(zr), c_ = zmap(zc);
        return (r >= 0 && c_ >= 0) ? Cg[r * d + c_] : 0.0f;
    };
    auto cz = [&](const float *cg, int zr) { const int r = zmap(zr); return r >= 0 ? cg[r] : 0.0f; };
    const float *Fg = a.F + (size_t)b * a.sF;
    const float *fg = a.f + (size_t)b * a.sf;
    const float *Cg = a.C + (size_t)b * a.sC;
    const float *cg = a.c + (size_t)b * a.sc;
    float *Kg = a.K + (size_t)b * a.sK;
    float *kg = a.k + (size_t)b * a.sk;
    int status = 0;

    if (BACKWARD) {
        // ---- resident operands ------------------------------------------------------
        float Fb0[4], Fb1[4];            // F~_c[4q+r][i]
        f32x4 Cd00, Cd11;                // C~[16a+4q+r][16c+i]
        f32x4 Cd01t;                     // rows 0..7: C_ux, row 8: c_x  (= (C~ tile (0,1))^T)
        f32x4 vterm;                     // c_x in lanes i == 8 (terminal v)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * q + r;         // state row of F~ / row of the x-part of C~
            const int ku = N + k;            // z index of the u-part rows (16..31; 16..23 exist)
            if (EXACT) {
                Fb0[r] = Fg[k * D + i];
                Fb1[r] = (i < M) ? Fg[k * D + N + i] : ((i == M) ? fg[k] : 0.0f);
                Cd00[r] = Cg[k * D + i];
                Cd01t[r] = (k < M) ? Cg[(N + k) * D + i] : ((k == M) ? cg[i] : 0.0f);     // C~[16+k][i] | c_x
                vterm[r] = (i == M) ? cg[k] : 0.0f;
                Cd11[r] = (ku < D) ? ((i < M) ? Cg[ku * D + N + i] : ((i == M) ? cg[ku] : 0.0f)) : 0.0f;
            } else {
                Fb0[r] = Fxx(Fg, k, i);
                Fb1[r] = (i < M) ? Fxu(Fg, k, i) : ((i == M && k < n) ? fg[k] : 0.0f);
                Cd00[r] = Czz(Cg, k, i);
                Cd01t[r] = (k < M) ? Czz(Cg, N + k, i) : ((k == M) ? cz(cg, i) : 0.0f);
                vterm[r] = (i == M) ? cz(cg, k) : 0.0f;
                float c11 = 0.0f;
                if (ku < D) {
                    if (i < M) c11 = (k >= m && i == k) ? 1.0f : Czz(Cg, ku, N + i);   // unit diagonal on padded actions
                    else if (i == M) c11 = cz(cg, ku);
                }
                Cd11[r] = c11;
            }
        }
        ConstFrag Fc0{}, Fc1{};                  // bf16x3 fragments of F~ (BF3 only)
        if (BF3) {
            Fc0 = const_frag(f32x4{Fb0[0], Fb0[1], Fb0[2], Fb0[3]});
            Fc1 = const_frag(f32x4{Fb1[0], Fb1[1], Fb1[2], Fb1[3]});
        }
        // terminal value function V = C_xx, v = c_x (lqr.py:67-68): v lives in lanes i == 8
        f32x4 Vd = Cd00, vd = vterm;
        float cst = 0.0f;
        int min_pivot_bits = 0x3f800000;     // smallest pivot seen, as float bits (int order == float order for >= 0)
        for (int idx = lane; idx < kZs; idx += kWave) lds[idx] = 0.0f;   // pad columns stay 0
        constexpr int kZero = kMs + 25 * 8;      // columns 25..31 of the elimination input are never written
        constexpr int kQx = kMs + 28 * 8;        // q_x staging in pad columns 28, 29
        const int t01_src = (i == M) ? kQx + 4 * q : kZero;
        const int g1_src = (i == M) ? kKs + (N + M) * 8 + q : kZero + q;
        __syncthreads();

        for (int t = T - 1; t >= 0; --t) {
            // 1. W = V F~ (+ v on column 24)                                  lqr.py:74,77-78
            f32x4 W0 = {0.f, 0.f, 0.f, 0.f}, W1 = {0.f, 0.f, 0.f, 0.f};
            if (BF3) {
                const VarFrag Vf = var_frag(Vd);
                W0 = mm_var_const(Vf, Fc0, W0);
                W1 = mm_var_const(Vf, Fc1, W1);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    W0 = mfma(Vd[r], Fb0[r], W0);
                    W1 = mfma(Vd[r], Fb1[r], W1);
                }
            }
            float fw = 0.0f, fv = 0.0f;
            if (VALUE) {     // f^T (V f) and f^T v for the const recursion (lqr.py:120)
                float pw = 0.0f, pv = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pw = fmaf(Fb1[r], W1[r], pw);
                    pv = fmaf(Fb1[r], vd[r], pv);
                }
                fw = wave_sum(i == M ? pw : 0.0f);
                fv = wave_sum(i == M ? pv : 0.0f);
            }
            W1 += vd;                            // vd is zero outside lanes i == 8 (see step 4)
            // 2. Q~ = C~ + F~^T W                                              lqr.py:75-78
            // Three tiles: Q_xx = F_x^T W_0; [Q_uu | q_u] = F~_1^T W_1; and W_1^T F_x, whose rows
            // 0..7 are Q_ux (= F_u^T V F_x: V is kept exactly symmetric, step 4, so this equals the
            // F~_1^T W_0 tile that is no longer computed) and whose row 8 is q_x^T = (V f + v)^T F_x.
            f32x4 T00 = Cd00, T01t = Cd01t, T11 = Cd11;
            if (BF3) {
                const VarFrag W0f = var_frag(W0), W1f = var_frag(W1);
                T00 = mm_const_var(Fc0, W0f, T00);
                T01t = mm_var_const(W1f, Fc0, T01t);
                T11 = mm_const_var(Fc1, W1f, T11);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    T00 = mfma(Fb0[r], W0[r], T00);
                    T01t = mfma(W1[r], Fb0[r], T01t);
                    T11 = mfma(Fb1[r], W1[r], T11);
                }
            }
            // 3. [Q_ux | Q_uu | q_u] -> column-per-lane layout through LDS; Q_xu and q_x staged
            if (q < 2) {
                *reinterpret_cast<f32x4 *>(&lds[kMs + i * 8 + 4 * q]) = T01t;                    // Q_ux[4q+r][i]
                if (i <= M) *reinterpret_cast<f32x4 *>(&lds[kMs + (N + i) * 8 + 4 * q]) = T11;
            } else if (q == 2) {
                lds[kQx + i] = T01t[0];                                                          // q_x[i]
            }
            lds_sync();
            // rows (2k, 2k+1) share a register pair so that one v_pk_fma_f32 updates both
            f32x2 M2[4];
            {
                const int c = lane & 31;
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(&lds[kMs + c * 8]);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(&lds[kMs + c * 8 + 4]);
                M2[0] = f32x2{lo[0], lo[1]}; M2[1] = f32x2{lo[2], lo[3]};
                M2[2] = f32x2{hi[0], hi[1]}; M2[3] = f32x2{hi[2], hi[3]};
            }
            float quk = 0.0f;
            float qu_saved[8];
            if (VALUE) {
#pragma unroll
                for (int p = 0; p < 8; ++p) qu_saved[p] = readlane(M2[p >> 1][p & 1], 24);
            }
            // K~ = -Q_uu^-1 [Q_ux | . | q_u]   (lqr.py:84-87; LDL^T on the upper triangle, wave_ldlt8.h)
            float Mr[8];
            ldlt8_solve_neg(M2, Mr, min_pivot_bits);
            if (VALUE) {
#pragma unroll
                for (int p = 0; p < 8; ++p) quk = fmaf(readlane(Mr[p], 24), qu_saved[p], quk);   // k^T q_u
            }
            // K~: columns 0..15 = K, column 24 = k
            if (lane < 32) {
                f32x4 lo, hi;
#pragma unroll
                for (int r = 0; r < 4; ++r) { lo[r] = Mr[r]; hi[r] = Mr[4 + r]; }
                *reinterpret_cast<f32x4 *>(&lds[kKs + lane * 8]) = lo;
                *reinterpret_cast<f32x4 *>(&lds[kKs + lane * 8 + 4]) = hi;
            }
            lds_sync();
            // 4. V' = Q_xx + Q_xu K ; v' = q_x + Q_xu k (column 24)            lqr.py:97-105
            //    contraction over the 8 actions as 2 k-steps: a = 4s + q
            //    vacc accumulates v' on q_x in column 24 (lanes i == 8); every other lane reads its
            //    operands from the always-zero pad columns, so the next v is 0 there.
            f32x4 vacc = *reinterpret_cast<const f32x4 *>(&lds[t01_src]);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float ax = lds[kMs + i * 8 + 4 * s2 + q];            // Q_xu[i][4s+q] = Q_ux[4s+q][i]
                const float g0 = lds[kKs + i * 8 + 4 * s2 + q];            // K[4s+q][i]
                const float g1 = lds[g1_src + 4 * s2];                     // K~[4s+q][24] in lanes i == 8
                T00 = mfma(ax, g0, T00);
                vacc = mfma(ax, g1, vacc);
            }
            // V' <- (V' + V'^T) / 2 (transpose through LDS).  The elimination above reads only the
            // upper triangle of Q_uu; that is consistent only while V carries no antisymmetric part
            // (otherwise the part it ignores, F_u^T a F_u, is missing from the closed-loop product
            // and the rounding-level asymmetry of V grows like |F_u K|^2 per step).
            {
                float *vt = &lds[kZs];                    // rollout buffer, idle during the sweep
                *reinterpret_cast<f32x4 *>(&vt[i * kVtLd + 4 * q]) = T00;
                lds_sync();
#pragma unroll
                for (int r = 0; r < 4; ++r) Vd[r] = 0.5f * (T00[r] + vt[(4 * q + r) * kVtLd + i]);
            }
            vd = vacc;
            // gains to HBM, row-major K[t][a][j], k[t][a] (the public layout)
            {
                const int ka = lane >> 3, jc = lane & 7;
                float2 kv;
                kv.x = lds[kKs + (2 * jc) * 8 + ka];
                kv.y = lds[kKs + (2 * jc + 1) * 8 + ka];
                if (EXACT) {
                    *reinterpret_cast<float2 *>(&Kg[(size_t)t * (M * N) + 2 * lane]) = kv;
                    if (lane < M) kg[(size_t)t * M + lane] = lds[kKs + 24 * 8 + lane];
                } else {
                    if (ka < m && 2 * jc < n) Kg[(size_t)t * m * n + ka * n + 2 * jc] = kv.x;
                    if (ka < m && 2 * jc + 1 < n) Kg[(size_t)t * m * n + ka * n + 2 * jc + 1] = kv.y;
                    if (lane < m) kg[(size_t)t * m + lane] = lds[kKs + 24 * 8 + lane];
                }
                if (OUT16 && a.K16) {                      // 16-bit copy of the policy (the rollout reads the fp32 gains)
                    uint16_t *Ko = a.K16 + ((size_t)b * T + t) * (m * n);
                    if (ka < m && 2 * jc < n) Ko[ka * n + 2 * jc] = lqr_to_bf16(kv.x);
                    if (ka < m && 2 * jc + 1 < n) Ko[ka * n + 2 * jc + 1] = lqr_to_bf16(kv.y);
                }
                if (OUT16 && a.k16 && lane < m) a.k16[((size_t)b * T + t) * m + lane] = lqr_to_bf16(lds[kKs + 24 * 8 + lane]);
            }
            if (VALUE) {
                // const += 1/2 k^T Q_uu k + k^T q_u + 1/2 f^T V f + f^T v with Q_uu k = -q_u
                // (lqr.py:113-121); f^T(V f) and f^T v were taken before v entered W.
                cst += 0.5f * quk + 0.5f * fw + fv;
                if (a.V) {
                    float *Vo = a.V + ((size_t)b * T + t) * (n * n);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (EXACT || (4 * q + r < n && i < n)) Vo[(4 * q + r) * n + i] = Vd[r];
                }
                if (a.v && i == M) {
                    float *vo = a.v + ((size_t)b * T + t) * n;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (EXACT || 4 * q + r < n) vo[4 * q + r] = vd[r];
                }
                if (a.cst && lane == 0) a.cst[(size_t)b * T + t] = cst;
                if (OUT16 && a.V16) {
                    uint16_t *Vo = a.V16 + ((size_t)b * T + t) * (n * n);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (EXACT || (4 * q + r < n && i < n)) Vo[(4 * q + r) * n + i] = lqr_to_bf16(Vd[r]);
                }
                if (OUT16 && a.v16 && i == M) {
                    uint16_t *vo = a.v16 + ((size_t)b * T + t) * n;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (EXACT || 4 * q + r < n) vo[4 * q + r] = lqr_to_bf16(vd[r]);
                }
                if (OUT16 && a.cst16 && lane == 0) a.cst16[(size_t)b * T + t] = lqr_to_bf16(cst);
            }
            lds_sync();
        }
        if (min_pivot_bits <= 0) status |= (min_pivot_bits == 0) ? TFMPC_ST_SINGULAR : TFMPC_ST_NOT_PD;
        if (VALUE && !(cst == cst)) status |= TFMPC_ST_NAN;
    }

    if (FORWARD) {
        // ---- resident operands of the rollout ------------------------------------------
        const int fi = lane >> 2, fc = lane & 3;       // F: row fi, columns 6fc..6fc+5
        const int ka = lane >> 3, jc = lane & 7;       // K: row ka, columns 2jc, 2jc+1
        float Fr[6];
        if (EXACT) {
            const float2 *p = reinterpret_cast<const float2 *>(Fg + fi * D + 6 * fc);
#pragma unroll
            for (int j = 0; j < 3; ++j) { const float2 v = p[j]; Fr[2 * j] = v.x; Fr[2 * j + 1] = v.y; }
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int zc = 6 * fc + j;
                Fr[j] = zc < N ? Fxx(Fg, fi, zc) : Fxu(Fg, fi, zc - N);
            }
        }
        const float f_part = (fc == 0 && (EXACT || fi < n)) ? fg[fi] : 0.0f;     // f enters one of the four partial sums
        // cost post-pass operands: A = C (2 row tiles x 6 k-steps, k = 4s + q), c in D layout
        float Ca0[6], Ca1[6];
        f32x4 cq0, cq1;
#pragma unroll
        for (int s2 = 0; s2 < 6; ++s2) {
            if (EXACT) {
                Ca0[s2] = Cg[i * D + 4 * s2 + q];
                Ca1[s2] = (i < M) ? Cg[(N + i) * D + 4 * s2 + q] : 0.0f;
            } else {
                Ca0[s2] = Czz(Cg, i, 4 * s2 + q);
                Ca1[s2] = (i < M) ? Czz(Cg, N + i, 4 * s2 + q) : 0.0f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cq0[r] = EXACT ? cg[4 * q + r] : cz(cg, 4 * q + r);
            cq1[r] = (q < 2) ? (EXACT ? cg[N + 4 * q + r] : cz(cg, N + 4 * q + r)) : 0.0f;
        }
        float *xs = a.states + (size_t)b * (T + 1) * n;
        float *us = a.actions + (size_t)b * T * m;
        float *cs = a.costs + (size_t)b * (T + 1);
        float *zs = &lds[kZs];
        __syncthreads();                               // gains written above are visible
        if (lane < N) {
            const float x = (EXACT || lane < n) ? a.x0[(size_t)b * n + lane] : 0.0f;
            zs[lane] = x;
            if (EXACT || lane < n) xs[lane] = x;
        }
        // gains of step t for this lane: K[ka][2jc], K[ka][2jc+1], k[ka]
        auto load_gain = [&](int t, float2 &Kv, float &kv) {
            if (EXACT) {
                Kv = *reinterpret_cast<const float2 *>(&Kg[(size_t)t * (M * N) + 2 * lane]);
                kv = kg[(size_t)t * M + ka];
            } else {
                const bool row = ka < m;
                Kv.x = (row && 2 * jc < n) ? Kg[(size_t)t * m * n + ka * n + 2 * jc] : 0.0f;
                Kv.y = (row && 2 * jc + 1 < n) ? Kg[(size_t)t * m * n + ka * n + 2 * jc + 1] : 0.0f;
                kv = row ? kg[(size_t)t * m + ka] : 0.0f;
            }
        };
        // Register ring kRing steps deep, statically indexed through the unrolled inner loop below (round 6): a rotation through copies
        // at the loop's back edge makes the compiler wait for the load it has just issued -- one memory round trip per step.
        float2 KR[kRing];
        float kR[kRing];
#pragma unroll
        for (int d = 0; d < kRing; ++d) {
            KR[d] = float2{0.f, 0.f};
            kR[d] = 0.0f;
            if (T > 0) load_gain(d < T ? d : T - 1, KR[d], kR[d]);
        }
        __syncthreads();

        // costs of rows [0, rows) of the chunk buffer: 1/2 z^T C z + c^T z  (lqr.py:41-47)
        // as C Z on the matrix cores, 16 timesteps per tile.
        auto chunk_costs = [&](int rows, float *out) {
            for (int nt = 0; nt * 16 < rows; ++nt) {
                const int row = (16 * nt + i < rows) ? 16 * nt + i : rows - 1;    // stay inside the chunk
                const float *zrow = zs + row * kZld;
                f32x4 D0 = {0.f, 0.f, 0.f, 0.f}, D1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < 6; ++s2) {
                    const float bz = zrow[4 * s2 + q];
                    D0 = mfma(Ca0[s2], bz, D0);
                    D1 = mfma(Ca1[s2], bz, D1);
                }
                float part = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    part = fmaf(zrow[4 * q + r], fmaf(0.5f, D0[r], cq0[r]), part);
                    if (q < 2) part = fmaf(zrow[N + 4 * q + r], fmaf(0.5f, D1[r], cq1[r]), part);
                }
                part += __shfl_xor(part, 16, kWave);
                part += __shfl_xor(part, 32, kWave);
                if (q == 0 && 16 * nt + i < rows) out[16 * nt + i] = part;
            }
        };

        for (int t0 = 0; t0 < T; t0 += kTC) {
            const int tc = (T - t0 < kTC) ? (T - t0) : kTC;
            for (int tb = 0; tb < tc; tb += kRing) {
#pragma unroll
                for (int d = 0; d < kRing; ++d) {
                    const int tt = tb + d;
                    if (tt >= tc) break;
                    const int t = t0 + tt;
                    float *zt = zs + tt * kZld;
                    const float2 Kc = KR[d];
                    const float kc = kR[d];
                    // this slot's next step -- UNCONDITIONAL (clamped: the last turns reload the final step): behind a branch the compiler
                    // can no longer count the loads in flight and waits for all of them
                    load_gain(t + kRing < T ? t + kRing : T - 1, KR[d], kR[d]);
                    // u = K x + k                                              lqr.py:143
                    const float2 xv = *reinterpret_cast<const float2 *>(&zt[2 * jc]);
                    float u = fmaf(Kc.x, xv.x, Kc.y * xv.y);
                    u += dpp<kDppXor1>(u);
                    u += dpp<kDppXor2>(u);
                    u += dpp<kDppHalfMirror>(u);
                    u += kc;
                    zt[N + ka] = u;                      // all eight lanes of the row hold the same sum
                    lds_sync();
                    // x' = F z + f                                              lqr.py:36-39
                    float xn = f_part;
                    const float2 *zp = reinterpret_cast<const float2 *>(&zt[6 * fc]);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float2 z2 = zp[j];
                        xn = fmaf(Fr[2 * j], z2.x, xn);
                        xn = fmaf(Fr[2 * j + 1], z2.y, xn);
                    }
                    xn += dpp<kDppXor1>(xn);
                    xn += dpp<kDppXor2>(xn);
                    zt[kZld + fi] = xn;                  // likewise: the four lanes of row fi agree
                    lds_sync();
                }
            }
            // chunk epilogue: stage costs on the matrix cores, bulk coalesced stores
            chunk_costs(tc, cs + t0);
            if (EXACT) {
                for (int idx = lane; idx < tc * N; idx += kWave)
                    xs[(size_t)(t0 + 1) * N + idx] = zs[(1 + idx / N) * kZld + (idx & (N - 1))];
                for (int idx = lane; idx < tc * M; idx += kWave)
                    us[(size_t)t0 * M + idx] = zs[(idx / M) * kZld + N + (idx & (M - 1))];
            } else {
                for (int idx = lane; idx < tc * n; idx += kWave)
                    xs[(size_t)(t0 + 1) * n + idx] = zs[(1 + idx / n) * kZld + idx % n];
                for (int idx = lane; idx < tc * m; idx += kWave)
                    us[(size_t)t0 * m + idx] = zs[(idx / m) * kZld + N + idx % m];
            }
            lds_sync();
            if (lane < N) zs[lane] = zs[tc * kZld + lane];      // carry x into row 0 of the next chunk
            lds_sync();
        }
        // final cost 1/2 x^T C_xx x + c_x^T x == stage cost with u = 0      lqr.py:49-57
        if (lane < M) zs[N + lane] = 0.0f;
        lds_sync();
        chunk_costs(1, cs + T);
        lds_sync();
        if (lane == 0) {
            const float fcost = cs[T];
            if (!(fcost == fcost)) status |= TFMPC_ST_NAN;
        }
    }

    if (a.status && lane == 0) a.status[b] = status;
}

// TFMPC_LQR_MFMA=f32 keeps the two big products of the sweep on the f32 MFMA; the default
// evaluates them as bf16x3 on the bf16 matrix cores (same fp32-level accuracy, see split3).
bool use_bf16x3()
{
    return !option_is(kOptLqrMfma, "f32");
}

// Register budget of a launch without value outputs: five waves per SIMD (see the note above the kernel); TFMPC_LQR_WAVES=4|5 forces
// (A/B timing, the bit-identity test).
int pick_eu()
{
    const int forced = option_int(kOptLqrWaves, 0);
    return forced == 4 ? 4 : 5;
}

template <bool BW, bool FW, bool VAL, bool O16, int EU>
int launch_eu(const LqrArgs &a, hipStream_t stream)
{
    const bool exact = a.n == N && a.m == M;
    const bool bf3 = BW && use_bf16x3();
    const dim3 grid(a.B), block(kWave);
    if (exact && bf3) hipLaunchKernelGGL((lqr_mfma16x8_kernel<BW, FW, VAL, true, true, O16, EU>), grid, block, 0, stream, a);
    else if (exact) hipLaunchKernelGGL((lqr_mfma16x8_kernel<BW, FW, VAL, true, false, O16, EU>), grid, block, 0, stream, a);
    else if (bf3) hipLaunchKernelGGL((lqr_mfma16x8_kernel<BW, FW, VAL, false, true, O16, EU>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((lqr_mfma16x8_kernel<BW, FW, VAL, false, false, O16, EU>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}

template <bool BW, bool FW, bool VAL, bool O16 = false>
int launch(const LqrArgs &a, hipStream_t stream)
{
    // the VALUE instantiations would spill at the budget for five (round 4): they keep four
    if constexpr (!VAL) {
        if (pick_eu() == 5) return launch_eu<BW, FW, VAL, O16, 5>(a, stream);
    }
    return launch_eu<BW, FW, VAL, O16, 4>(a, stream);
}

}  // namespace

// Exact headline shape, or any smaller shape that is still worth a 16 x 8 tile grid (below
// n + m = 7 the lane-per-instance kernel of lqr_lane.hip takes over).
bool lqr_mfma_supported(int n, int m) { return n >= 1 && m >= 1 && n <= N && m <= M && n + m > 6; }

int lqr_mfma_launch(const LqrArgs &a, bool backward, bool forward, hipStream_t stream)
{
    if (a.K16 || a.k16 || a.V16 || a.v16 || a.cst16) {       // 16-bit outputs: the value-function code is compiled in
        if (backward && forward) return launch<true, true, true, true>(a, stream);
        if (backward) return launch<true, false, true, true>(a, stream);
    }
    const bool value = a.V || a.v || a.cst;
    if (backward && forward) return value ? launch<true, true, true>(a, stream) : launch<true, true, false>(a, stream);
    if (backward) return value ? launch<true, false, true>(a, stream) : launch<true, false, false>(a, stream);
    return launch<false, true, false>(a, stream);
}

}  // namespace tfmpc
