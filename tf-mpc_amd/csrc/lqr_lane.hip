// lqr_lane.hip -- LQR (tfmpc/solvers/lqr.py:59-166) for tiny problems: ONE LANE per problem
// instance, everything in registers, 64 instances per wavefront.  Used for n + m <= 6
// (navlin n = m = 2 of BASELINE configs[1], the README's n = 3, m = 2, ...) when the batch
// is large enough to fill lanes; the arithmetic follows the reference's operation order
// exactly like lqr_generic.hip (general inverse with row pivoting, four-term V / v update,
// const recursion).  Regime: HBM / launch-latency bound, so the point of
// this mapping is 64 x more instances in flight per wave than wave-per-instance.
#include <hip/hip_runtime.h>

#include "lqr_kernels.h"
#include "small_linalg.h"

namespace tfmpc {

using small::Mat;

template <int N, int M, bool BACKWARD, bool FORWARD>
__global__ __launch_bounds__(64) void lqr_lane_kernel(LqrArgs a)
{
    constexpr int D = N + M;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const int T = a.T;

    Mat<N, D> F;
    Mat<D, D> C;
    float f[N], c[D];
    {
        const float *Fg = a.F + (size_t)b * a.sF, *Cg = a.C + (size_t)b * a.sC;
        const float *fg = a.f + (size_t)b * a.sf, *cg = a.c + (size_t)b * a.sc;
#pragma unroll
        for (int i = 0; i < N * D; ++i) F.a[i] = Fg[i];
#pragma unroll
        for (int i = 0; i < D * D; ++i) C.a[i] = Cg[i];
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = fg[i];
#pragma unroll
        for (int i = 0; i < D; ++i) c[i] = cg[i];
    }
    float *Kg = a.K + (size_t)b * a.sK;
    float *kg = a.k + (size_t)b * a.sk;
    int status = 0;

    if (BACKWARD) {
        Mat<N, N> V;
        float v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {                      // lqr.py:67-68
            v[i] = c[i];
#pragma unroll
            for (int j = 0; j < N; ++j) V(i, j) = C(i, j);
        }
        float cst = 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            const Mat<D, N> W = small::mul_tn<N, D, N>(F, V);                 // F^T V          :74
            Mat<D, D> Q;
            float q[D];
#pragma unroll
            for (int r = 0; r < D; ++r) {                                     // :75-78
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    s1 = fmaf(W(r, k), f[k], s1);
                    s2 = fmaf(F(k, r), v[k], s2);
                }
                q[r] = c[r] + s1 + s2;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    float s = C(r, j);
#pragma unroll
                    for (int k = 0; k < N; ++k) s = fmaf(W(r, k), F(k, j), s);
                    Q(r, j) = s;
                }
            }
            Mat<M, M + 1 + N> aug;                                            // :84-87
#pragma unroll
            for (int r = 0; r < M; ++r) {
#pragma unroll
                for (int j = 0; j < M; ++j) aug(r, j) = Q(N + r, N + j);
                aug(r, M) = q[N + r];
#pragma unroll
                for (int j = 0; j < N; ++j) aug(r, M + 1 + j) = Q(N + r, j);
            }
            if (small::gauss_jordan<M, 1 + N, true>(aug)) status |= TFMPC_ST_SINGULAR;
            Mat<M, N> K;
            float k[M];
#pragma unroll
            for (int r = 0; r < M; ++r) {
                k[r] = -aug(r, M);
                kg[(size_t)t * M + r] = k[r];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    K(r, j) = -aug(r, M + 1 + j);
                    Kg[(size_t)t * M * N + r * N + j] = K(r, j);
                }
            }
            Mat<N, M> KtQ;                                                    // K^T Q_uu       :95
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    float s = 0.0f;
#pragma unroll
                    for (int r = 0; r < M; ++r) s = fmaf(K(r, i), Q(N + r, N + j), s);
                    KtQ(i, j) = s;
                }
            Mat<N, N> Vn;
            float vn[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {                                     // :97-105
                float t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
#pragma unroll
                for (int r = 0; r < M; ++r) {
                    t1 = fmaf(Q(i, N + r), k[r], t1);
                    t2 = fmaf(K(r, i), q[N + r], t2);
                    t3 = fmaf(KtQ(i, r), k[r], t3);
                }
                vn[i] = q[i] + t1 + t2 + t3;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
                    for (int r = 0; r < M; ++r) {
                        s1 = fmaf(Q(i, N + r), K(r, j), s1);
                        s2 = fmaf(K(r, i), Q(N + r, j), s2);
                        s3 = fmaf(KtQ(i, r), K(r, j), s3);
                    }
                    Vn(i, j) = Q(i, j) + s1 + s2 + s3;
                }
            }
            float part = 0.0f;                                                // :113-121
#pragma unroll
            for (int r = 0; r < M; ++r) {
                float quk = 0.0f;
#pragma unroll
                for (int j = 0; j < M; ++j) quk = fmaf(Q(N + r, N + j), k[j], quk);
                part += k[r] * (0.5f * quk + q[N + r]);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float vf = 0.0f;
#pragma unroll
                for (int j = 0; j < N; ++j) vf = fmaf(V(i, j), f[j], vf);
                part += f[i] * (0.5f * vf + v[i]);
            }
            cst += part;
            V = Vn;
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = vn[i];
            if (a.V) {
#pragma unroll
                for (int i = 0; i < N * N; ++i) a.V[((size_t)b * T + t) * N * N + i] = V.a[i];
            }
            if (a.v) {
#pragma unroll
                for (int i = 0; i < N; ++i) a.v[((size_t)b * T + t) * N + i] = v[i];
            }
            if (a.cst) a.cst[(size_t)b * T + t] = cst;
        }
        if (!(cst == cst)) status |= TFMPC_ST_NAN;
    }

    if (FORWARD) {
        float *xs = a.states + (size_t)b * (T + 1) * N;
        float *us = a.actions + (size_t)b * T * M;
        float *cs = a.costs + (size_t)b * (T + 1);
        float z[D];
#pragma unroll
        for (int i = 0; i < N; ++i) { z[i] = a.x0[(size_t)b * N + i]; xs[i] = z[i]; }
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int r = 0; r < M; ++r) {                                     // u = K x + k    :143
                float u = kg[(size_t)t * M + r];
#pragma unroll
                for (int j = 0; j < N; ++j) u = fmaf(Kg[(size_t)t * M * N + r * N + j], z[j], u);
                z[N + r] = u;
                us[(size_t)t * M + r] = u;
            }
            float cost = 0.0f;                                                // :41-47
#pragma unroll
            for (int r = 0; r < D; ++r) {
                float cz = 0.0f;
#pragma unroll
                for (int j = 0; j < D; ++j) cz = fmaf(C(r, j), z[j], cz);
                cost += z[r] * (0.5f * cz + c[r]);
            }
            cs[t] = cost;
            float xn[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {                                     // :36-39
                float x = f[i];
#pragma unroll
                for (int j = 0; j < D; ++j) x = fmaf(F(i, j), z[j], x);
                xn[i] = x;
            }
#pragma unroll
            for (int i = 0; i < N; ++i) { z[i] = xn[i]; xs[(size_t)(t + 1) * N + i] = xn[i]; }
        }
        float fc = 0.0f;                                                      // :49-57
#pragma unroll
        for (int r = 0; r < N; ++r) {
            float cz = 0.0f;
#pragma unroll
            for (int j = 0; j < N; ++j) cz = fmaf(C(r, j), z[j], cz);
            fc += z[r] * (0.5f * cz + c[r]);
        }
        cs[T] = fc;
        if (!(fc == fc)) status |= TFMPC_ST_NAN;
    }
    if (a.status) a.status[b] = status;
}

template <int N, int M>
static int launch_nm(const LqrArgs &a, bool bw, bool fw, hipStream_t stream)
{
    const dim3 grid((a.B + 63) / 64), block(64);
    if (bw && fw) hipLaunchKernelGGL((lqr_lane_kernel<N, M, true, true>), grid, block, 0, stream, a);
    else if (bw) hipLaunchKernelGGL((lqr_lane_kernel<N, M, true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((lqr_lane_kernel<N, M, false, true>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}

bool lqr_lane_supported(int n, int m)
{
    return (n == 1 && m == 1) || (n == 2 && m == 1) || (n == 2 && m == 2) || (n == 3 && m == 2) ||
           (n == 3 && m == 3) || (n == 4 && m == 2);
}

int lqr_lane_launch(const LqrArgs &a, bool bw, bool fw, hipStream_t stream)
{
#define TFMPC_LANE_CASE(N_, M_) if (a.n == N_ && a.m == M_) return launch_nm<N_, M_>(a, bw, fw, stream)
    TFMPC_LANE_CASE(1, 1);
    TFMPC_LANE_CASE(2, 1);
    TFMPC_LANE_CASE(2, 2);
    TFMPC_LANE_CASE(3, 2);
    TFMPC_LANE_CASE(3, 3);
    TFMPC_LANE_CASE(4, 2);
#undef TFMPC_LANE_CASE
    return TFMPC_ERR_UNSUPPORTED;
}

}  // namespace tfmpc
