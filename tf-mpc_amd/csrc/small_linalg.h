// small_linalg.h -- fixed-size dense helpers for the lane-per-instance kernels: when
// n, m <= 4 a whole problem instance fits one lane's registers, so 64 instances run per
// wavefront with no cross-lane traffic at all (HBM / latency-bound regime of the navigation
// configs).  Every loop has compile-time bounds and unrolls completely;
// arrays are indexed with compile-time constants only, so they stay in VGPRs.
#pragma once

#include <hip/hip_runtime.h>

namespace tfmpc {
namespace small {

template <int R, int C>
struct Mat {
    float a[R * C];
    __device__ __forceinline__ float &operator()(int i, int j) { return a[i * C + j]; }
    __device__ __forceinline__ float operator()(int i, int j) const { return a[i * C + j]; }
};

template <int R, int C>
__device__ __forceinline__ Mat<R, C> zeros()
{
    Mat<R, C> m;
#pragma unroll
    for (int i = 0; i < R * C; ++i) m.a[i] = 0.0f;
    return m;
}

// C = A^T B  (A: K x R, B: K x C)
template <int K, int R, int C>
__device__ __forceinline__ Mat<R, C> mul_tn(const Mat<K, R> &A, const Mat<K, C> &B)
{
    Mat<R, C> out;
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) s = fmaf(A(k, i), B(k, j), s);
            out(i, j) = s;
        }
    return out;
}

// C = A B  (A: R x K, B: K x C)
template <int R, int K, int C>
__device__ __forceinline__ Mat<R, C> mul_nn(const Mat<R, K> &A, const Mat<K, C> &B)
{
    Mat<R, C> out;
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) s = fmaf(A(i, k), B(k, j), s);
            out(i, j) = s;
        }
    return out;
}

// In-place Gauss-Jordan on [A | RHS] (R rows, R + W columns).  PIVOT: partial pivoting
// (general inverse, lqr.py:84); otherwise no pivoting and the return value tells whether
// every pivot was positive (the Cholesky success test of ilqr.py:358).  `active[r]`
// false marks identity rows (box-QP clamped dimensions).  Returns 0 ok, 1 bad pivot.
template <int R, int W, bool PIVOT>
__device__ __forceinline__ int gauss_jordan(Mat<R, R + W> &A)
{
    int bad = 0;
#pragma unroll
    for (int p = 0; p < R; ++p) {
        if (PIVOT) {
            // bring the largest |A(i,p)|, i >= p, to row p (branch-free swaps)
#pragma unroll
            for (int i = p + 1; i < R; ++i) {
                const bool sw = fabsf(A(i, p)) > fabsf(A(p, p));
#pragma unroll
                for (int j = 0; j < R + W; ++j) {
                    const float x = A(p, j), y = A(i, j);
                    A(p, j) = sw ? y : x;
                    A(i, j) = sw ? x : y;
                }
            }
        }
        const float pv = A(p, p);
        if (PIVOT ? (pv == 0.0f) : !(pv > 0.0f)) bad = 1;
        const float inv = 1.0f / pv;
#pragma unroll
        for (int j = 0; j < R + W; ++j) A(p, j) *= inv;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (i == p) continue;
            const float f = A(i, p);
#pragma unroll
            for (int j = 0; j < R + W; ++j) A(i, j) = fmaf(-f, A(p, j), A(i, j));
        }
    }
    return bad;
}

}  // namespace small
}  // namespace tfmpc
