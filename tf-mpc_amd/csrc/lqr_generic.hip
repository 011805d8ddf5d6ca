// lqr_generic.hip -- shape-generic batched LQR on gfx950: one wavefront per problem
// instance, every operand of the instance staged in that wave's LDS slice.
//
// Replaces tfmpc/solvers/lqr.py of the reference: LQR.backward (:59-129),
// LQR.forward (:131-161), LQR.solve (:163-166).  This is the variant that accepts
// any (n, m) whose tiles fit one wave's LDS; the dispatcher in lqr_dispatch.hip
// routes the BASELINE.json headline shape (n=16, m=8) to the MFMA kernel instead.
//
// Data layout in HBM (tfmpc_hip.h): batch-major, row-major fp32.  A wave reads its
// instance's F, f, C, c once (contiguous, coalesced), keeps them resident in LDS
// for all T steps, streams K_t / k_t (and optionally V_t, v_t, const_t) out during
// the backward sweep and back in during the rollout.
#include <hip/hip_runtime.h>

#include "lqr_kernels.h"
#include "wave_ops.h"

namespace tfmpc {

struct LqrSmem {
    int ldd, ldn, ldm, lda, width;
    float *F, *f, *C, *c, *V, *v, *W, *Q, *q, *aug, *fac, *prow, *K, *k, *KtQ, *Vn, *vn, *z, *xn;
};

__host__ __device__ inline size_t lqr_smem_floats(int n, int m)
{
    const int d = n + m;
    const int ldd = odd_ld(d), ldn = odd_ld(n), ldm = odd_ld(m), width = m + 1 + n, lda = odd_ld(width);
    size_t s = 0;
    s += (size_t)n * ldd + n;        // F, f
    s += (size_t)d * ldd + d;        // C, c
    s += (size_t)n * ldn + n;        // V, v
    s += (size_t)d * ldn;            // W = F^T V
    s += (size_t)d * ldd + d;        // Q, q
    s += (size_t)m * lda + m + width;  // aug, fac, prow
    s += (size_t)m * ldn + m;        // K, k
    s += (size_t)n * ldm;            // K^T Q_uu
    s += (size_t)n * ldn + n;        // Vn, vn
    s += (size_t)d + n;              // z, xn
    return s;
}

__device__ inline LqrSmem lqr_carve(float *base, int n, int m)
{
    LqrSmem s;
    const int d = n + m;
    s.ldd = odd_ld(d);
    s.ldn = odd_ld(n);
    s.ldm = odd_ld(m);
    s.width = m + 1 + n;
    s.lda = odd_ld(s.width);
    float *p = base;
    s.F = p; p += n * s.ldd;
    s.f = p; p += n;
    s.C = p; p += d * s.ldd;
    s.c = p; p += d;
    s.V = p; p += n * s.ldn;
    s.v = p; p += n;
    s.W = p; p += d * s.ldn;
    s.Q = p; p += d * s.ldd;
    s.q = p; p += d;
    s.aug = p; p += m * s.lda;
    s.fac = p; p += m;
    s.prow = p; p += s.width;
    s.K = p; p += m * s.ldn;
    s.k = p; p += m;
    s.KtQ = p; p += n * s.ldm;
    s.Vn = p; p += n * s.ldn;
    s.vn = p; p += n;
    s.z = p; p += d;
    s.xn = p; p += n;
    return s;
}

template <bool BACKWARD, bool FORWARD>
__global__ __launch_bounds__(kWave) void lqr_generic_kernel(LqrArgs a)
{
    extern __shared__ float smem[];
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int n = a.n, m = a.m, d = n + m, T = a.T;
    LqrSmem s = lqr_carve(smem, n, m);
    const int ldd = s.ldd, ldn = s.ldn, ldm = s.ldm, lda = s.lda;

    load_matrix(s.F, ldd, a.F + (size_t)b * a.sF, n, d);
    load_matrix(s.C, ldd, a.C + (size_t)b * a.sC, d, d);
    for (int i = lane; i < n; i += kWave) s.f[i] = a.f[(size_t)b * a.sf + i];
    for (int i = lane; i < d; i += kWave) s.c[i] = a.c[(size_t)b * a.sc + i];
    wsync();

    int status = 0;
    float *Kg = a.K ? a.K + (size_t)b * a.sK : nullptr;
    float *kg = a.k ? a.k + (size_t)b * a.sk : nullptr;

    if (BACKWARD) {
        // terminal condition V = C_xx, v = c_x, const = 0              (lqr.py:67-69)
        wave_for_2d(n, n, [&](int i, int j, int) { s.V[i * ldn + j] = s.C[i * ldd + j]; });
        for (int i = lane; i < n; i += kWave) s.v[i] = s.c[i];
        float cst = 0.0f;
        wsync();

        for (int t = T - 1; t >= 0; --t) {
            // W = F^T V  [d][n]                                        (lqr.py:74)
            wave_matmul_mfma(d, n, n,
                        [&](int r, int k) { return s.F[k * ldd + r]; },
                        [&](int k, int j) { return s.V[k * ldn + j]; },
                        [](int, int) { return 0.0f; },
                        [&](int r, int j, float x) { s.W[r * ldn + j] = x; });
            wsync();
            // Q = C + W F ; q = c + W f + F^T v                        (lqr.py:75-78)
            wave_matmul_mfma(d, d, n,
                        [&](int r, int k) { return s.W[r * ldn + k]; },
                        [&](int k, int j) { return s.F[k * ldd + j]; },
                        [&](int r, int j) { return s.C[r * ldd + j]; },
                        [&](int r, int j, float x) { s.Q[r * ldd + j] = x; });
            for (int r = lane; r < d; r += kWave) {
                float s1 = 0.0f, s2 = 0.0f;
                for (int k = 0; k < n; ++k) {
                    s1 = fmaf(s.W[r * ldn + k], s.f[k], s1);
                    s2 = fmaf(s.F[k * ldd + r], s.v[k], s2);
                }
                s.q[r] = s.c[r] + s1 + s2;
            }
            wsync();
            // [Q_uu | q_u | Q_ux] -> Gauss-Jordan -> [I | Q_uu^-1 q_u | Q_uu^-1 Q_ux]
            // (general inverse with row pivoting, lqr.py:84-87)
            wave_for_2d(m, s.width, [&](int r, int j, int) {
                float x;
                if (j < m) x = s.Q[(n + r) * ldd + n + j];
                else if (j == m) x = s.q[n + r];
                else x = s.Q[(n + r) * ldd + (j - m - 1)];
                s.aug[r * lda + j] = x;
            });
            wsync();
            if (wave_gauss_jordan<true>(s.aug, lda, m, s.width, s.fac, s.prow)) status |= TFMPC_ST_SINGULAR;
            wave_for_2d(m, n, [&](int r, int j, int idx) {
                const float x = -s.aug[r * lda + m + 1 + j];
                s.K[r * ldn + j] = x;
                if (Kg) Kg[(size_t)t * m * n + idx] = x;
                if (a.K16) a.K16[((size_t)b * T + t) * m * n + idx] = lqr_to_bf16(x);
            });
            for (int r = lane; r < m; r += kWave) {
                const float x = -s.aug[r * lda + m];
                s.k[r] = x;
                if (kg) kg[(size_t)t * m + r] = x;
                if (a.k16) a.k16[((size_t)b * T + t) * m + r] = lqr_to_bf16(x);
            }
            wsync();
            // K^T Q_uu  [n][m]                                         (lqr.py:95)
            wave_matmul_mfma(n, m, m,
                        [&](int i, int k) { return s.K[k * ldn + i]; },
                        [&](int k, int j) { return s.Q[(n + k) * ldd + n + j]; },
                        [](int, int) { return 0.0f; },
                        [&](int i, int j, float x) { s.KtQ[i * ldm + j] = x; });
            wsync();
            // V' = Q_xx + Q_xu K + K^T Q_ux + K^T Q_uu K               (lqr.py:97-100)
            // as ONE product [Q_xu | K^T | K^T Q_uu] [K ; Q_ux ; K] accumulated onto Q_xx
            wave_matmul_mfma(n, n, 3 * m,
                        [&](int i, int k) {
                            return k < m ? s.Q[i * ldd + n + k] : (k < 2 * m ? s.K[(k - m) * ldn + i] : s.KtQ[i * ldm + (k - 2 * m)]);
                        },
                        [&](int k, int j) {
                            return k < m ? s.K[k * ldn + j] : (k < 2 * m ? s.Q[(n + k - m) * ldd + j] : s.K[(k - 2 * m) * ldn + j]);
                        },
                        [&](int i, int j) { return s.Q[i * ldd + j]; },
                        [&](int i, int j, float x) { s.Vn[i * ldn + j] = x; });
            // v' = q_x + Q_xu k + K^T q_u + K^T Q_uu k                 (lqr.py:102-105)
            for (int i = lane; i < n; i += kWave) {
                float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
                for (int k = 0; k < m; ++k) {
                    s1 = fmaf(s.Q[i * ldd + n + k], s.k[k], s1);
                    s2 = fmaf(s.K[k * ldn + i], s.q[n + k], s2);
                    s3 = fmaf(s.KtQ[i * ldm + k], s.k[k], s3);
                }
                s.vn[i] = s.q[i] + s1 + s2 + s3;
            }
            // const += 1/2 k^T Q_uu k + k^T q_u + 1/2 f^T V f + f^T v  (lqr.py:113-121,
            // with the value function BEFORE this step's update)
            float part = 0.0f;
            for (int r = lane; r < m; r += kWave) {
                float quk = 0.0f;
                for (int k = 0; k < m; ++k) quk = fmaf(s.Q[(n + r) * ldd + n + k], s.k[k], quk);
                part += s.k[r] * (0.5f * quk + s.q[n + r]);
            }
            for (int i = lane; i < n; i += kWave) {
                float vf = 0.0f;
                for (int k = 0; k < n; ++k) vf = fmaf(s.V[i * ldn + k], s.f[k], vf);
                part += s.f[i] * (0.5f * vf + s.v[i]);
            }
            cst += wave_sum(part);
            wsync();
            wave_for_2d(n, n, [&](int i, int j, int idx) {
                const float x = s.Vn[i * ldn + j];
                s.V[i * ldn + j] = x;
                if (a.V) a.V[((size_t)b * T + t) * n * n + idx] = x;
                if (a.V16) a.V16[((size_t)b * T + t) * n * n + idx] = lqr_to_bf16(x);
            });
            for (int i = lane; i < n; i += kWave) {
                const float x = s.vn[i];
                s.v[i] = x;
                if (a.v) a.v[((size_t)b * T + t) * n + i] = x;
                if (a.v16) a.v16[((size_t)b * T + t) * n + i] = lqr_to_bf16(x);
            }
            if (a.cst && lane == 0) a.cst[(size_t)b * T + t] = cst;
            if (a.cst16 && lane == 0) a.cst16[(size_t)b * T + t] = lqr_to_bf16(cst);
            wsync();
        }

        if (!(cst == cst)) status |= TFMPC_ST_NAN;
    }

    if (FORWARD) {
        const float *Kr = a.K + (size_t)b * a.sK;
        const float *kr = a.k + (size_t)b * a.sk;
        float *xs = a.states + (size_t)b * (T + 1) * n;
        float *us = a.actions + (size_t)b * T * m;
        float *cs = a.costs + (size_t)b * (T + 1);
        for (int i = lane; i < n; i += kWave) {
            const float x = a.x0[(size_t)b * n + i];
            s.z[i] = x;
            xs[i] = x;
        }
        float last_cost = 0.0f;
        for (int t = 0; t < T; ++t) {
            load_matrix(s.K, ldn, Kr + (size_t)t * m * n, m, n);
            for (int r = lane; r < m; r += kWave) s.k[r] = kr[(size_t)t * m + r];
            wsync();
            for (int r = lane; r < m; r += kWave) {        // u = K x + k      (lqr.py:143)
                float u = s.k[r];
                for (int j = 0; j < n; ++j) u = fmaf(s.K[r * ldn + j], s.z[j], u);
                s.z[n + r] = u;
                us[(size_t)t * m + r] = u;
            }
            wsync();
            float part = 0.0f;                             // 1/2 z^T C z + c^T z (lqr.py:41-47)
            for (int r = lane; r < d; r += kWave) {
                float cz = 0.0f;
                for (int j = 0; j < d; ++j) cz = fmaf(s.C[r * ldd + j], s.z[j], cz);
                part += s.z[r] * (0.5f * cz + s.c[r]);
            }
            for (int i = lane; i < n; i += kWave) {        // x' = F z + f     (lqr.py:36-39)
                float x = s.f[i];
                for (int j = 0; j < d; ++j) x = fmaf(s.F[i * ldd + j], s.z[j], x);
                s.xn[i] = x;
            }
            const float cost = wave_sum(part);
            if (lane == 0) cs[t] = cost;
            wsync();
            for (int i = lane; i < n; i += kWave) {
                const float x = s.xn[i];
                s.z[i] = x;
                xs[(size_t)(t + 1) * n + i] = x;
            }
            wsync();
        }
        float part = 0.0f;                                 // final cost      (lqr.py:49-57)
        for (int r = lane; r < n; r += kWave) {
            float cz = 0.0f;
            for (int j = 0; j < n; ++j) cz = fmaf(s.C[r * ldd + j], s.z[j], cz);
            part += s.z[r] * (0.5f * cz + s.c[r]);
        }
        last_cost = wave_sum(part);
        if (lane == 0) cs[T] = last_cost;
        if (!(last_cost == last_cost)) status |= TFMPC_ST_NAN;
    }

    if (a.status && lane == 0) a.status[b] = status;
}

size_t lqr_generic_smem_bytes(int n, int m) { return lqr_smem_floats(n, m) * sizeof(float); }

template <bool BW, bool FW>
static int launch(const LqrArgs &a, hipStream_t stream)
{
    const size_t smem = lqr_generic_smem_bytes(a.n, a.m);
    if (smem > kMaxLdsBytes) return TFMPC_ERR_UNSUPPORTED;
    auto kern = lqr_generic_kernel<BW, FW>;
    if (smem > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return TFMPC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(kWave), smem, stream, a);
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}

int lqr_generic_launch(const LqrArgs &a, bool backward, bool forward, hipStream_t stream)
{
    if (backward && forward) return launch<true, true>(a, stream);
    if (backward) return launch<true, false>(a, stream);
    return launch<false, true>(a, stream);
}

}  // namespace tfmpc
