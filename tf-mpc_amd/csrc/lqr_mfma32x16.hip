// lqr_mfma32x16.hip -- LQR backward + forward (tfmpc/solvers/lqr.py:59-166) for shapes beyond the 16 x 8 tile of
// lqr_mfma16x8.hip, up to n = 32, m = 16 (BASELINE configs[4]'s literal dims), on the gfx950 matrix cores.  One
// wavefront owns one problem instance; smaller shapes run zero-padded into the 32 x 16 tile grid (unit diagonal on
// the padded part of C_uu, so the elimination stays regular and the padded gains are exactly 0).
//
// The sweep is the one of lqr_mfma16x8.hip written for 2 x 2 tiles of 16 x 16:
//   F~ = [F_x | F_u | f 0..] is 32 x 64 = 2 x 4 tiles, V is 2 x 2 tiles kept EXACTLY symmetric, every product is
//   "bf16x3" on the bf16 matrix cores (mfma_bf16x3.h: fp32 operands split into three bf16 parts, fp32 accumulation).
//   1. W = V F~ (2 x 4 tiles, contraction over 2 k-tiles); column 48 of W is V f, += v.
//   2. Q_xx = F_x^T W_x;  Q_ux = W_u^T F_x (rows = actions);  q_x = F_x^T (V f + v) (a column);
//      [Q_uu | q_u] = F_u^T [W_u | V f + v]:  108 v_mfma_f32_16x16x32_bf16 per step, fed from registers -- the
//      accumulator layout of a tile IS the operand layout of the next product (k = 4 q + r), as in the small kernel.
//   3. [Q_ux | Q_uu | q_u] (16 x 49) crosses LDS once into "one column per lane, 16 rows in registers" and is solved by
//      the no-pivot LDL^T of wave_ldlt.h (upper triangle of Q_uu only) -> K~ = -Q_uu^-1 [Q_ux | q_u].
//   4. V' = Q_xx + Q_xu K, v' = q_x + Q_xu k on the f32 matrix cores (contraction over the 16 actions), then
//      V' <- (V' + V'^T) / 2 through an LDS transpose (lqr.py:97-105 in Schur form; see lqr_mfma16x8.hip step 4).
//   K_t, k_t stream to HBM row-major (the public layout) for the rollout.
// Forward rollout (lqr.py:141-155): u = K x + k and x' = F z + f as wave-wide FMA mat-vecs (4 lanes per gain row, 2 per
// row of F), z_t staged in LDS; stage costs priced afterwards as C Z on the f32 matrix cores.
//
// PRECONDITION as for every fast LQR kernel (include/tfmpc_hip.h): C symmetric, C_uu > 0.
#include <hip/hip_runtime.h>

#include "lqr_kernels.h"
#include "mfma_bf16x3.h"
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

// per-wave LDS slice (floats); the rollout's z buffer reuses the sweep's staging areas
constexpr int kMld = 64;                    // elimination input, row-major [16 actions][64 cols]: Q_ux | Q_uu | q_u | 0..
constexpr int kMs = 0;
constexpr int kKs = kMs + M * kMld;         // gains row-major [16][32] = the public K layout
constexpr int kkv = kKs + M * N;            // k[16]
constexpr int kVt = kkv + 16;               // V' transpose staging [32][33]
constexpr int kVld = 33;
constexpr int kCst = kVt + N * kVld;        // constants: c_x[32], c_u[16], 16 zeros
constexpr int kZeros = kCst + 48;
constexpr int kSweepFloats = kZeros + 16;
constexpr int kZld = 52;                    // rollout rows z_t = [x(32); u(16)] (52: rows 16-byte aligned, banks spread)
constexpr int kTC = 48;                     // timesteps per rollout chunk
constexpr int kLdsFloats = (kSweepFloats > (kTC + 1) * kZld ? kSweepFloats : (kTC + 1) * kZld) + 8;

struct Tile2x2 { f32x4 t[2][2]; };

template <bool BACKWARD, bool FORWARD, bool VALUE, bool OUT16 = false>      // OUT16: see lqr_mfma16x8.hip
__global__ __launch_bounds__(kWave, 2) void lqr_mfma32x16_kernel(LqrArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int i = lane & 15, q = lane >> 4;
    const int T = a.T;
    const int n = a.n, m = a.m, d = n + m;
    const float *Fg = a.F + (size_t)b * a.sF;
    const float *fg = a.f + (size_t)b * a.sf;
    const float *Cg = a.C + (size_t)b * a.sC;
    const float *cg = a.c + (size_t)b * a.sc;
    float *Kg = a.K + (size_t)b * a.sK;
    float *kg = a.k + (size_t)b * a.sk;
    // padded-index accessors: x index in [0,32), u index in [0,16); z index in [0,48): 0..31 -> x, 32..47 -> u
    auto zmap = [&](int zi) { return zi < N ? (zi < n ? zi : -1) : (zi - N < m ? n + zi - N : -1); };
    auto Fz = [&](int row, int zi) { const int c_ = zmap(zi); return (row < n && c_ >= 0) ? Fg[row * d + c_] : 0.0f; };
    auto Czz = [&](int zr, int zc) {
        const int r = zmap(zr), c_ = zmap(zc);
        if (r >= 0 && c_ >= 0) return Cg[r * d + c_];
        return (zr == zc && zr >= N + m && zr < D) ? 1.0f : 0.0f;          // unit diagonal on padded actions
    };
    auto cz = [&](int zr) { const int r = zmap(zr); return r >= 0 ? cg[r] : 0.0f; };
    int status = 0;

    if (BACKWARD) {
        // ---- resident operands: bf16x3 fragments of F~ (tile (kt, ct): rows 16 kt + 4 q + r, column 16 ct + i) -----
        ConstFrag Fc[2][4];
        float fcol[2][4];                                 // f[16 kt + 4 q + r] in lanes i == 0 (const recursion)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * kt + 4 * q + r, col = 16 * ct + i;
                    v[r] = col < D ? Fz(row, col) : ((col == D && row < n) ? fg[row] : 0.0f);
                }
                Fc[kt][ct] = const_frag(v);
                if (ct == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) fcol[kt][r] = v[r];
                }
            }
        }
        f32x4 Cxx[2][2], Cux[2], Cuu;                     // accumulator initialisers (rows 4 q + r of the tile, column i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 2; ++b_) Cxx[a_][b_][r] = Czz(16 * a_ + 4 * q + r, 16 * b_ + i);
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) Cux[b_][r] = Czz(N + 4 * q + r, 16 * b_ + i);
            Cuu[r] = Czz(N + 4 * q + r, N + i);
        }
        for (int idx = lane; idx < kSweepFloats; idx += kWave) lds[idx] = 0.0f;
        __syncthreads();
        if (lane < 48) lds[kCst + lane] = cz(lane);       // c_x[0..31], c_u[0..15]
        __syncthreads();
        // columns of the affine slot: lanes i == 0 read the vector, all others a block of zeros
        const int cx_src0 = (i == 0) ? kCst + 4 * q : kZeros, cx_src1 = (i == 0) ? kCst + 16 + 4 * q : kZeros;
        const int cu_src = (i == 0) ? kCst + 32 + 4 * q : kZeros;
        const int kv_src = (i == 0) ? kkv + q : kZeros + q;              // k[4 s + q] for the v' product (lanes i == 0)

        // terminal value function V = C_xx, v = c_x (lqr.py:67-68): v lives in column 0 of the affine tile
        f32x4 Vd[2][2], vd[2];
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) Vd[a_][b_] = Cxx[a_][b_];
        vd[0] = *reinterpret_cast<const f32x4 *>(&lds[cx_src0]);
        vd[1] = *reinterpret_cast<const f32x4 *>(&lds[cx_src1]);
        float cst = 0.0f;
        int min_pivot_bits = 0x3f800000;

        for (int t = T - 1; t >= 0; --t) {
            // 1. W = V F~ (+ v on column 48)                                           lqr.py:74,77-78
            // block (rt, kt) of the symmetric V as A operand = the tile Vd[kt][rt] read "transposed" (k = 4 q + r)
            f32x4 W[2][4];
            {
                VarFrag Vf[2][2];
#pragma unroll
                for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                    for (int b_ = 0; b_ < 2; ++b_) Vf[a_][b_] = var_frag(Vd[a_][b_]);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) acc = mm_var_const(Vf[kt][rt], Fc[kt][ct], acc);
                        W[rt][ct] = acc;
                    }
            }
            float fw = 0.0f, fv = 0.0f;
            if (VALUE) {     // f^T (V f) and f^T v for the const recursion (lqr.py:120), before v enters W
                float pw = 0.0f, pv = 0.0f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pw = fmaf(fcol[kt][r], W[kt][3][r], pw);
                        pv = fmaf(fcol[kt][r], vd[kt][r], pv);
                    }
                fw = wave_sum(i == 0 ? pw : 0.0f);
                fv = wave_sum(i == 0 ? pv : 0.0f);
            }
            W[0][3] += vd[0];
            W[1][3] += vd[1];
            // 2. Q~ = C~ + F~^T W                                                       lqr.py:75-78
            f32x4 Qxx[2][2], Qux[2], qx[2], Quu, qu;
            {
                VarFrag Wf[2][4];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) Wf[kt][ct] = var_frag(W[kt][ct]);
#pragma unroll
                for (int a_ = 0; a_ < 2; ++a_) {
#pragma unroll
                    for (int b_ = 0; b_ < 2; ++b_) {
                        f32x4 acc = Cxx[a_][b_];                                   // Q_xx = C_xx + F_x^T W_x
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) acc = mm_const_var(Fc[kt][a_], Wf[kt][b_], acc);
                        Qxx[a_][b_] = acc;
                    }
                    f32x4 acc = *reinterpret_cast<const f32x4 *>(&lds[a_ == 0 ? cx_src0 : cx_src1]);   // q_x = c_x + F_x^T (V f + v)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) acc = mm_const_var(Fc[kt][a_], Wf[kt][3], acc);
                    qx[a_] = acc;
                }
#pragma unroll
                for (int b_ = 0; b_ < 2; ++b_) {
                    f32x4 acc = Cux[b_];                                           // Q_ux = C_ux + W_u^T F_x (V symmetric)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) acc = mm_var_const(Wf[kt][2], Fc[kt][b_], acc);
                    Qux[b_] = acc;
                }
                f32x4 acc = Cuu;                                                   // Q_uu = C_uu + F_u^T W_u
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) acc = mm_const_var(Fc[kt][2], Wf[kt][2], acc);
                Quu = acc;
                acc = *reinterpret_cast<const f32x4 *>(&lds[cu_src]);              // q_u = c_u + F_u^T (V f + v)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) acc = mm_const_var(Fc[kt][2], Wf[kt][3], acc);
                qu = acc;
            }
            // 3. [Q_ux | Q_uu | q_u] -> row-major staging -> one column per lane
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lds[kMs + (4 * q + r) * kMld + i] = Qux[0][r];
                lds[kMs + (4 * q + r) * kMld + 16 + i] = Qux[1][r];
                lds[kMs + (4 * q + r) * kMld + N + i] = Quu[r];
                if (i == 0) lds[kMs + (4 * q + r) * kMld + N + M] = qu[r];
            }
            __syncthreads();
            f32x2 M2[M / 2];
#pragma unroll
            for (int e = 0; e < M / 2; ++e)
                M2[e] = f32x2{lds[kMs + (2 * e) * kMld + lane], lds[kMs + (2 * e + 1) * kMld + lane]};
            float quk = 0.0f;
            float qu_saved[M];
            if (VALUE) {
#pragma unroll
                for (int p = 0; p < M; ++p) qu_saved[p] = readlane(M2[p >> 1][p & 1], N + M);
            }
            // K~ = -Q_uu^-1 [Q_ux | . | q_u]   (lqr.py:84-87; LDL^T on the upper triangle, wave_ldlt.h)
            float Mr[M];
            ldlt_solve_neg<M, N>(M2, Mr, min_pivot_bits);
            if (VALUE) {
#pragma unroll
                for (int p = 0; p < M; ++p) quk = fmaf(readlane(Mr[p], N + M), qu_saved[p], quk);   // k^T q_u
            }
            // gains row-major into LDS: lane c < 32 holds column c of K, lane 48 holds k
            if (lane < N) {
#pragma unroll
                for (int e = 0; e < M; ++e) lds[kKs + e * N + lane] = Mr[e];
            } else if (lane == N + M) {
#pragma unroll
                for (int e = 0; e < M; e += 4) *reinterpret_cast<f32x4 *>(&lds[kkv + e]) = f32x4{Mr[e], Mr[e + 1], Mr[e + 2], Mr[e + 3]};
            }
            __syncthreads();
            // 4. V' = Q_xx + Q_xu K ; v' = q_x + Q_xu k                                 lqr.py:97-105
            //    contraction over the 16 actions as 4 k-steps: action = 4 s + q
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const float ax0 = lds[kMs + (4 * s2 + q) * kMld + i];              // Q_xu[i][4s+q] = Q_ux[4s+q][i]
                const float ax1 = lds[kMs + (4 * s2 + q) * kMld + 16 + i];
                const float g0 = lds[kKs + (4 * s2 + q) * N + i];                  // K[4s+q][i]
                const float g1 = lds[kKs + (4 * s2 + q) * N + 16 + i];
                const float gk = lds[kv_src + 4 * s2];                             // k[4s+q] in lanes i == 0
                Qxx[0][0] = mfma(ax0, g0, Qxx[0][0]);
                Qxx[0][1] = mfma(ax0, g1, Qxx[0][1]);
                Qxx[1][0] = mfma(ax1, g0, Qxx[1][0]);
                Qxx[1][1] = mfma(ax1, g1, Qxx[1][1]);
                qx[0] = mfma(ax0, gk, qx[0]);
                qx[1] = mfma(ax1, gk, qx[1]);
            }
            // V' <- (V' + V'^T) / 2 (transpose through LDS): the sweep uses the symmetry of V (step 2, and the
            // elimination reads the upper triangle of Q_uu only), so no antisymmetric rounding residue may survive a step
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
                    for (int r = 0; r < 4; ++r)
                        Vd[a_][b_][r] = 0.5f * (Qxx[a_][b_][r] + lds[kVt + (16 * b_ + i) * kVld + 16 * a_ + 4 * q + r]);
            vd[0] = qx[0];
            vd[1] = qx[1];
            // gains to HBM, row-major K[t][a][j], k[t][a] (the public layout)
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
            if (OUT16 && a.K16) {                          // 16-bit copy of the policy (the rollout reads the fp32 gains)
                uint16_t *Ko = a.K16 + ((size_t)b * T + t) * (m * n);
                for (int idx = lane; idx < M * N; idx += kWave) {
                    const int ka = idx >> 5, j = idx & 31;
                    if (ka < m && j < n) Ko[ka * n + j] = lqr_to_bf16(lds[kKs + idx]);
                }
            }
            if (OUT16 && a.k16 && lane < m) a.k16[((size_t)b * T + t) * m + lane] = lqr_to_bf16(lds[kkv + lane]);
            if (VALUE) {
                // const += 1/2 k^T Q_uu k + k^T q_u + 1/2 f^T V f + f^T v with Q_uu k = -q_u (lqr.py:113-121)
                cst += 0.5f * quk + 0.5f * fw + fv;
                if (a.V) {
                    float *Vo = a.V + ((size_t)b * T + t) * (n * n);
#pragma unroll
                    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * a_ + 4 * q + r, col = 16 * b_ + i;
                                if (row < n && col < n) Vo[row * n + col] = Vd[a_][b_][r];
                            }
                }
                if (a.v && i == 0) {
                    float *vo = a.v + ((size_t)b * T + t) * n;
#pragma unroll
                    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * a_ + 4 * q + r < n) vo[16 * a_ + 4 * q + r] = vd[a_][r];
                }
                if (a.cst && lane == 0) a.cst[(size_t)b * T + t] = cst;
                if (OUT16 && a.V16) {
                    uint16_t *Vo = a.V16 + ((size_t)b * T + t) * (n * n);
#pragma unroll
                    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * a_ + 4 * q + r, col = 16 * b_ + i;
                                if (row < n && col < n) Vo[row * n + col] = lqr_to_bf16(Vd[a_][b_][r]);
                            }
                }
                if (OUT16 && a.v16 && i == 0) {
                    uint16_t *vo = a.v16 + ((size_t)b * T + t) * n;
#pragma unroll
                    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * a_ + 4 * q + r < n) vo[16 * a_ + 4 * q + r] = lqr_to_bf16(vd[a_][r]);
                }
                if (OUT16 && a.cst16 && lane == 0) a.cst16[(size_t)b * T + t] = lqr_to_bf16(cst);
            }
            __syncthreads();
        }
        if (min_pivot_bits <= 0) status |= (min_pivot_bits == 0) ? TFMPC_ST_SINGULAR : TFMPC_ST_NOT_PD;
        if (VALUE && !(cst == cst)) status |= TFMPC_ST_NAN;
    }

    if (FORWARD) {
        // ---- resident operands of the rollout (indices derived from an opaque copy of the lane id: loaded HERE, after the
        // sweep, not hoisted above it where 72 more live registers would spill the sweep) -----------------------------
        int lo_ = lane;
        asm volatile("" : "+v"(lo_));
        const int fi = lo_ >> 1, fc = lo_ & 1;         // F: row fi, z columns 24 fc .. 24 fc + 23
        const int ka = lo_ >> 2, jc = lo_ & 3;         // K: row ka, columns 8 jc .. 8 jc + 7
        const int io = lo_ & 15, qo = lo_ >> 4;
        float Fr[24];
#pragma unroll
        for (int j = 0; j < 24; ++j) Fr[j] = Fz(fi, 24 * fc + j);
        const float f_part = (fc == 0 && fi < n) ? fg[fi] : 0.0f;
        // cost post-pass operands: A = C (3 row tiles x 12 k-steps, k = 4 s + q), c in D layout
        float Ca[3][12];
        f32x4 cq[3];
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
#pragma unroll
            for (int s2 = 0; s2 < 12; ++s2) {
                const int zr = 16 * rt + io, zc = 4 * s2 + qo;
                const int r = zmap(zr), c_ = zmap(zc);
                Ca[rt][s2] = (r >= 0 && c_ >= 0) ? Cg[r * d + c_] : 0.0f;     // (no unit diagonal here: padded z entries are 0 anyway)
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) cq[rt][r] = cz(16 * rt + 4 * qo + r);
        }
        float *xs = a.states + (size_t)b * (T + 1) * n;
        float *us = a.actions + (size_t)b * T * m;
        float *cs = a.costs + (size_t)b * (T + 1);
        float *zs = &lds[0];
        __syncthreads();                               // the sweep's staging areas are free now
        if (lane < N) {
            const float x = lane < n ? a.x0[(size_t)b * n + lane] : 0.0f;
            zs[lane] = x;
            if (lane < n) xs[lane] = x;
        }
        // gains of step t for this lane: K[ka][8 jc .. 8 jc + 7], k[ka]
        auto load_gain = [&](int t, float (&Kv)[8], float &kv) {
            if (n == N && m == M) {
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(&Kg[(size_t)t * (M * N) + ka * N + 8 * jc]);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(&Kg[(size_t)t * (M * N) + ka * N + 8 * jc + 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { Kv[j] = lo[j]; Kv[4 + j] = hi[j]; }
                kv = kg[(size_t)t * M + ka];
            } else {
                const bool row = ka < m;
#pragma unroll
                for (int j = 0; j < 8; ++j) Kv[j] = (row && 8 * jc + j < n) ? Kg[(size_t)t * m * n + ka * n + 8 * jc + j] : 0.0f;
                kv = row ? kg[(size_t)t * m + ka] : 0.0f;
            }
        };
        float Kn[8], kn = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) Kn[j] = 0.0f;
        if (T > 0) load_gain(0, Kn, kn);
        __syncthreads();

        // costs of rows [0, rows) of the chunk buffer: 1/2 z^T C z + c^T z  (lqr.py:41-47) as C Z on the matrix cores,
        // 16 timesteps per tile
        auto chunk_costs = [&](int rows, float *out) {
            for (int nt = 0; nt * 16 < rows; ++nt) {
                const int row = (16 * nt + i < rows) ? 16 * nt + i : rows - 1;
                const float *zrow = zs + row * kZld;
                f32x4 Dz[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int s2 = 0; s2 < 12; ++s2) {
                    const float bz = zrow[4 * s2 + q];
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt) Dz[rt] = mfma(Ca[rt][s2], bz, Dz[rt]);
                }
                float part = 0.0f;
#pragma unroll
                for (int rt = 0; rt < 3; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part = fmaf(zrow[16 * rt + 4 * q + r], fmaf(0.5f, Dz[rt][r], cq[rt][r]), part);
                part += __shfl_xor(part, 16, kWave);
                part += __shfl_xor(part, 32, kWave);
                if (q == 0 && 16 * nt + i < rows) out[16 * nt + i] = part;
            }
        };

        for (int t0 = 0; t0 < T; t0 += kTC) {
            const int tc = (T - t0 < kTC) ? (T - t0) : kTC;
            for (int tt = 0; tt < tc; ++tt) {
                const int t = t0 + tt;
                float *zt = zs + tt * kZld;
                float Kc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) Kc[j] = Kn[j];
                const float kc = kn;
                if (t + 1 < T) load_gain(t + 1, Kn, kn);      // prefetch the next step's gains
                // u = K x + k                                              lqr.py:143
                const f32x4 xlo = *reinterpret_cast<const f32x4 *>(&zt[8 * jc]), xhi = *reinterpret_cast<const f32x4 *>(&zt[8 * jc + 4]);
                float u = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { u = fmaf(Kc[j], xlo[j], u); u = fmaf(Kc[4 + j], xhi[j], u); }
                u += dpp<kDppXor1>(u);
                u += dpp<kDppXor2>(u);
                u += kc;
                zt[N + ka] = u;                      // the four lanes of the row hold the same sum
                __syncthreads();
                // x' = F z + f                                              lqr.py:36-39
                float xn = f_part;
#pragma unroll
                for (int j4 = 0; j4 < 6; ++j4) {
                    const f32x4 z4 = *reinterpret_cast<const f32x4 *>(&zt[24 * fc + 4 * j4]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) xn = fmaf(Fr[4 * j4 + j], z4[j], xn);
                }
                xn += dpp<kDppXor1>(xn);
                zt[kZld + fi] = xn;                  // both lanes of row fi agree
                __syncthreads();
            }
            // chunk epilogue: stage costs on the matrix cores, bulk coalesced stores
            chunk_costs(tc, cs + t0);
            for (int idx = lane; idx < tc * n; idx += kWave) xs[(size_t)(t0 + 1) * n + idx] = zs[(1 + idx / n) * kZld + idx % n];
            for (int idx = lane; idx < tc * m; idx += kWave) us[(size_t)t0 * m + idx] = zs[(idx / m) * kZld + N + idx % m];
            __syncthreads();
            if (lane < N) zs[lane] = zs[tc * kZld + lane];      // carry x into row 0 of the next chunk
            __syncthreads();
        }
        // final cost 1/2 x^T C_xx x + c_x^T x == stage cost with u = 0      lqr.py:49-57
        if (lane < M) zs[N + lane] = 0.0f;
        __syncthreads();
        chunk_costs(1, cs + T);
        __syncthreads();
        if (lane == 0) {
            const float fcost = cs[T];
            if (!(fcost == fcost)) status |= TFMPC_ST_NAN;
        }
    }

    if (a.status && lane == 0) a.status[b] = status;
}

template <bool BW, bool FW, bool VAL, bool O16 = false>
int launch(const LqrArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL((lqr_mfma32x16_kernel<BW, FW, VAL, O16>), dim3(a.B), dim3(kWave), 0, stream, a);
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}

}  // namespace

// Shapes beyond the 16 x 8 tile that still fit the 32 x 16 one (and are not served better by the lane kernels).
bool lqr_mfma32_supported(int n, int m)
{
    return n >= 1 && m >= 1 && n <= N && m <= M && !(n <= 16 && m <= 8) && n + m >= 12;
}

int lqr_mfma32_launch(const LqrArgs &a, bool backward, bool forward, hipStream_t stream)
{
    if (a.K16 || a.k16 || a.V16 || a.v16 || a.cst16) {
        if (backward && forward) return launch<true, true, true, true>(a, stream);
        if (backward) return launch<true, false, true, true>(a, stream);
    }
    const bool value = a.V || a.v || a.cst;
    if (backward && forward) return value ? launch<true, true, true>(a, stream) : launch<true, true, false>(a, stream);
    if (backward) return value ? launch<true, false, true>(a, stream) : launch<true, false, false>(a, stream);
    return launch<false, true, false>(a, stream);
}

}  // namespace tfmpc
