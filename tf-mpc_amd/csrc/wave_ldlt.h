// wave_ldlt.h -- the R x R SPD solve of the large-tile matrix-core sweep (lqr_mfma32x16.hip), the generalisation of
// wave_ldlt8.h: -Q_uu^-1 [Q_ux | Q_uu | q_u] with the R-row system held "one column per lane, R rows in registers"
// (lanes KQ .. KQ+R-1 hold the columns of Q_uu).
#pragma once

#include <hip/hip_runtime.h>

#include "wave_ldlt8.h"

namespace tfmpc {

// LDL^T elimination without pivoting that reads only the upper triangle of Q_uu (row p, column KQ + s, p <= s), like
// the Cholesky it stands for (lqr.py:84-87 inverse; ilqr.py:357-362 Cholesky solve):
//   forward   row_s -= L[s][p] row_p (s > p),  L[s][p] = row_p[KQ+s] / d_p  by symmetry of the Schur complement;
//   backward  X_p = row_p / d_p - sum_{s>p} L[s][p] X_s.
// The multipliers -L[s][p] are wave-uniform (v_readlane).  With R = 16 there are 120 of them -- more than the scalar
// register file holds across the two sweeps -- so the backward sweep reads them again from an untouched copy of
// -row_p / d_p (Nkeep) instead of keeping them live.  Rows (2k, 2k+1) share a register pair (v_pk_fma_f32).
// min_pivot_bits tracks the smallest pivot as float bits: <= 0 at the end <=> Q_uu was not positive definite.
// CAUTION: ignoring the lower triangle is only consistent while the caller keeps the value matrix exactly symmetric.
// `last_ninv` (optional): receives -1 / d_{R-1}, the reciprocal of the LAST pivot with its sign folded in -- which is the (R-1, R-1) entry of
// -Q_uu^-1 (for an LDL^T with unit lower L the last diagonal entry of the inverse is 1 / d_last): the one entry the gain-reusing iLQR kernel
// cannot read off its fifteen identity columns (ilqr_lq_mfma32.hip).
template <int R, int KQ>
__device__ __forceinline__ void ldlt_solve_neg(f32x2 (&M2)[R / 2], float (&X)[R], int &min_pivot_bits, float *last_ninv = nullptr)
{
    static_assert(R % 2 == 0, "rows come in register pairs");
    f32x2 N2[R / 2];
    float Nkeep[R];
#pragma unroll
    for (int p = 0; p < R; ++p) {
        const int pp = p >> 1, ps = p & 1;
        const float Mp = M2[pp][ps];
        const int pvb = __builtin_amdgcn_readlane(__builtin_bit_cast(int, Mp), KQ + p);
        asm("s_min_i32 %0, %0, %1" : "+s"(min_pivot_bits) : "s"(pvb) : "scc");
        const float ninv = __builtin_amdgcn_rcpf(-__builtin_bit_cast(float, pvb));
        const float Mn = Mp * ninv;                      // -row_p / d_p
        if (last_ninv && p == R - 1) *last_ninv = ninv;
        N2[pp][ps] = Mn;
        Nkeep[p] = Mn;
        float nl[R];                                     // -L[s][p], s > p
#pragma unroll
        for (int s = p + 1; s < R; ++s)
            nl[s] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Mn), KQ + s));
        if (ps == 0) M2[pp][1] = fmaf(nl[p + 1], Mp, M2[pp][1]);
        const f32x2 Mpp = {Mp, Mp};
#pragma unroll
        for (int k = pp + 1; k < R / 2; ++k)
            M2[k] = __builtin_elementwise_fma(f32x2{nl[2 * k], nl[2 * k + 1]}, Mpp, M2[k]);
    }
#pragma unroll
    for (int s = R - 1; s >= 1; --s) {
        const float Ns = N2[s >> 1][s & 1];
        const f32x2 Nss = {Ns, Ns};
        float nl[R];                                     // -L[s][k], k < s: lane KQ + s of the untouched -row_k / d_k
#pragma unroll
        for (int k = 0; k < s; ++k)
            nl[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Nkeep[k]), KQ + s));
        if (s & 1) N2[s >> 1][0] = fmaf(nl[s - 1], Ns, N2[s >> 1][0]);
#pragma unroll
        for (int k = 0; k < (s >> 1); ++k)
            N2[k] = __builtin_elementwise_fma(f32x2{nl[2 * k], nl[2 * k + 1]}, Nss, N2[k]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) X[r] = N2[r >> 1][r & 1];
}

}  // namespace tfmpc
