// wave_ldlt8.h -- the 8 x 8 SPD solve of the matrix-core sweeps (lqr_mfma16x8.hip, ilqr_lq_mfma.hip):
// -Q_uu^-1 [Q_ux | q_u] with the 8 x 25 system held "one column per lane, eight rows in registers"
// (lane c < 16: column c of Q_ux; lanes 16..23: columns of Q_uu; lane 24: q_u).
#pragma once

#include <hip/hip_runtime.h>

namespace tfmpc {

using f32x2 = __attribute__((ext_vector_type(2))) float;

// LDL^T elimination that reads only the upper triangle of Q_uu (column 16+s of row p, p <= s), like
// the Cholesky it stands for (lqr.py:84-87 inverse; ilqr.py:357-362 Cholesky solve):
//   forward   row_s -= L[s][p] row_p (s > p),  L[s][p] = row_p[16+s] / d_p  by symmetry of the Schur
//             complement -- so ONE v_readlane per multiplier serves both sweeps (36 per solve; a
//             Gauss-Jordan on the full matrix needs 64);
//   backward  X_p = row_p / d_p - sum_{s>p} L[s][p] X_s.
// The multipliers -L[s][p] are wave-uniform scalars (SGPRs); the sign of the result is folded into the
// pivot reciprocal (v_rcp_f32, 1 ulp).  Rows (2k, 2k+1) share a register pair so one v_pk_fma_f32
// updates both.  min_pivot_bits tracks the smallest pivot as float bits on the scalar unit: <= 0 at
// the end <=> Q_uu was not positive definite (the Cholesky failure test of ilqr.py:358).
// CAUTION: ignoring the lower triangle is only consistent while the caller keeps the value matrix
// exactly symmetric (see the callers' symmetrisation step).
__device__ __forceinline__ void ldlt8_solve_neg(f32x2 (&M2)[4], float (&X)[8], int &min_pivot_bits)
{
    constexpr int kQuu = 16;
    float nl[8][8];
    f32x2 N2[4];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int pp = p >> 1, ps = p & 1;
        const float Mp = M2[pp][ps];
        const int pvb = __builtin_amdgcn_readlane(__builtin_bit_cast(int, Mp), kQuu + p);
        asm("s_min_i32 %0, %0, %1" : "+s"(min_pivot_bits) : "s"(pvb) : "scc");
        const float ninv = __builtin_amdgcn_rcpf(-__builtin_bit_cast(float, pvb));
        const float Mn = Mp * ninv;                      // -row_p / d_p
        N2[pp][ps] = Mn;
#pragma unroll
        for (int s = p + 1; s < 8; ++s)
            nl[s][p] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Mn), kQuu + s));
        if (ps == 0) M2[pp][1] = fmaf(nl[p + 1][p], Mp, M2[pp][1]);
        const f32x2 Mpp = {Mp, Mp};
#pragma unroll
        for (int k = pp + 1; k < 4; ++k) {
#ifdef TFMPC_LDLT_SCALAR_FMA        // A/B: two v_fma_f32 instead of one v_pk_fma_f32
            M2[k][0] = fmaf(nl[2 * k][p], Mp, M2[k][0]);
            M2[k][1] = fmaf(nl[2 * k + 1][p], Mp, M2[k][1]);
#else
            M2[k] = __builtin_elementwise_fma(f32x2{nl[2 * k][p], nl[2 * k + 1][p]}, Mpp, M2[k]);
#endif
        }
    }
#pragma unroll
    for (int s = 7; s >= 1; --s) {
        const float Ns = N2[s >> 1][s & 1];
        const f32x2 Nss = {Ns, Ns};
        if (s & 1) N2[s >> 1][0] = fmaf(nl[s][s - 1], Ns, N2[s >> 1][0]);
#pragma unroll
        for (int k = 0; k < (s >> 1); ++k) {
#ifdef TFMPC_LDLT_SCALAR_FMA
            N2[k][0] = fmaf(nl[s][2 * k], Ns, N2[k][0]);
            N2[k][1] = fmaf(nl[s][2 * k + 1], Ns, N2[k][1]);
#else
            N2[k] = __builtin_elementwise_fma(f32x2{nl[s][2 * k], nl[s][2 * k + 1]}, Nss, N2[k]);
#endif
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) X[r] = N2[r >> 1][r & 1];
}

}  // namespace tfmpc
