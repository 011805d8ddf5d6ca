// ilqr_core.h -- wave-cooperative device routines of the iLQR path: the regularised
// backward pass (tfmpc/solvers/ilqr.py:94-172), its three controllers (Cholesky-type
// solve :357-362, box-QP :364-387 + tfmpc/utils/optimization.py:6-101, bang-bang
// :140-141) and the closed-loop rollout (:174-212).  One wavefront owns one problem
// instance; all step-local matrices live in that wave's LDS slice (IlqrSmem).
#pragma once

#include <hip/hip_runtime.h>

#include "envs.h"
#include "wave_ops.h"

namespace tfmpc {

struct IlqrSmem {
    int n, m, ldn, ldm, lda, width;
    int bf16;                                        // 1: HBM stores of trajectories / gains round to bf16
    float *fx, *fu, *lx, *lu, *lxx, *luu, *lux;      // model at step t (lux holds l_xu^T)
    float *Vx, *Vxx, *W1, *W2;                       // value terms, f_x^T V_xx, f_u^T V_xx
    float *Qx, *Qu, *Qxx, *Quu, *Qux, *Quur, *Quxr;  // Q terms and their regularised twins
    float *aug, *fac, *prow;                         // Gauss-Jordan workspace
    float *K, *k, *KtQ;
    float *xv, *uv, *xn, *xh, *uh;                   // state/action scratch
    float *qx, *qg, *qs, *qc, *qlo, *qhi, *qfree, *qgc;   // box-QP vectors
};

// Value as it would read back from a bf16 slab (round to nearest even on the fp32 bits); identity
// in the default fp32-storage mode.  Only what goes to HBM is rounded, never the registers / LDS.
__device__ __forceinline__ float stq(const IlqrSmem &s, float v)
{
    if (!s.bf16) return v;
    unsigned u = __float_as_uint(v);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    return __uint_as_float(u);
}

__host__ __device__ inline size_t ilqr_smem_floats(int n, int m)
{
    const size_t ldn = odd_ld(n), ldm = odd_ld(m), width = m + 1 + n, lda = odd_ld((int)width);
    size_t s = 0;
    s += n * ldn + n * ldm + n + m + n * ldn + m * ldm + m * ldn;      // fx fu lx lu lxx luu lux
    s += n + n * ldn + n * ldn + m * ldn;                              // Vx Vxx W1 W2
    s += n + m + n * ldn + m * ldm + m * ldn + m * ldm + m * ldn;      // Qx Qu Qxx Quu Qux Quur Quxr
    s += m * lda + m + width;                                          // aug fac prow
    s += m * ldn + m + n * ldm;                                        // K k KtQ
    s += 3 * n + 2 * m + n;                                            // xv xn xh, uv uh (+pad)
    s += 8 * m;                                                        // box-QP vectors
    return s;
}

__device__ inline float *ilqr_carve(IlqrSmem &s, float *p, int n, int m)
{
    s.n = n; s.m = m; s.bf16 = 0;
    s.ldn = odd_ld(n); s.ldm = odd_ld(m); s.width = m + 1 + n; s.lda = odd_ld(s.width);
    const int ldn = s.ldn, ldm = s.ldm;
    s.fx = p; p += n * ldn;   s.fu = p; p += n * ldm;
    s.lx = p; p += n;         s.lu = p; p += m;
    s.lxx = p; p += n * ldn;  s.luu = p; p += m * ldm;  s.lux = p; p += m * ldn;
    s.Vx = p; p += n;         s.Vxx = p; p += n * ldn;
    s.W1 = p; p += n * ldn;   s.W2 = p; p += m * ldn;
    s.Qx = p; p += n;         s.Qu = p; p += m;
    s.Qxx = p; p += n * ldn;  s.Quu = p; p += m * ldm;  s.Qux = p; p += m * ldn;
    s.Quur = p; p += m * ldm; s.Quxr = p; p += m * ldn;
    s.aug = p; p += m * s.lda; s.fac = p; p += m; s.prow = p; p += s.width;
    s.K = p; p += m * ldn;    s.k = p; p += m;  s.KtQ = p; p += n * ldm;
    s.xv = p; p += n; s.xn = p; p += n; s.xh = p; p += n; p += n;
    s.uv = p; p += m; s.uh = p; p += m;
    s.qx = p; p += m; s.qg = p; p += m; s.qs = p; p += m; s.qc = p; p += m;
    s.qlo = p; p += m; s.qhi = p; p += m; s.qfree = p; p += m; s.qgc = p; p += m;
    return p;
}

// Reduced LDS slice for envs whose cost Hessians vanish identically (HVAC, Reservoir):
// the backward pass is then the adjoint recursion below and needs no n x n Q terms.
__host__ __device__ inline size_t ilqr_adjoint_smem_floats(int n, int m)
{
    return 6 * (size_t)n + 5 * (size_t)m;        // vectors only: f_x, f_u are never materialised, K == 0
}

__device__ inline float *ilqr_carve_adjoint(IlqrSmem &s, float *p, int n, int m)
{
    s = IlqrSmem{};
    s.n = n; s.m = m; s.bf16 = 0;
    s.ldn = odd_ld(n); s.ldm = odd_ld(m); s.width = m + 1 + n; s.lda = odd_ld(s.width);
    s.lx = p; p += n;  s.Vx = p; p += n;  s.Qx = p; p += n;
    s.xv = p; p += n;  s.xn = p; p += n;  s.xh = p; p += n;
    s.lu = p; p += m;  s.Qu = p; p += m;  s.k = p; p += m;  s.uv = p; p += m;  s.uh = p; p += m;
    return p;
}

// ------------------------------------------------------------------ box-QP ----
// 1/2 x^T H x + q^T x for x in LDS (optimization.py:8-11).
__device__ inline float qp_objective(const float *H, int ld, const float *q, const float *x, int m)
{
    float part = 0.0f;
    for (int i = lane_id(); i < m; i += kWave) {
        float hx = 0.0f;
        for (int j = 0; j < m; ++j) hx = fmaf(H[i * ld + j], x[j], hx);
        part += x[i] * (0.5f * hx + q[i]);
    }
    return wave_sum(part);
}

// Builds [H~ | rhs~] in s.aug where rows/cols of clamped dimensions are replaced by
// identity rows (so the elimination solves the free sub-system only) and eliminates.
// rhs(i, c) supplies nrhs right-hand sides.  Returns non-zero if H_ff is not PD.
template <class FRhs>
__device__ inline int qp_solve_free(IlqrSmem &s, const float *H, int ldh, const float *freem, int nrhs, FRhs rhs)
{
    const int m = s.m, lda = s.lda, width = m + nrhs;
    for (int idx = lane_id(); idx < m * width; idx += kWave) {
        const int i = idx / width, j = idx - i * width;
        const bool fi = freem[i] != 0.0f;
        float v;
        if (j < m) v = (fi && freem[j] != 0.0f) ? H[i * ldh + j] : ((i == j) ? 1.0f : 0.0f);
        else v = fi ? rhs(i, j - m) : 0.0f;
        s.aug[i * lda + j] = v;
    }
    wsync();
    return wave_gauss_jordan<false>(s.aug, lda, m, width, s.fac, s.prow);
}

// projected_newton_qp (optimization.py:6-101).  H (m x m, leading dim ldh), q, bounds
// s.qlo/s.qhi and the start point s.qx are in LDS; on return s.qx holds the solution
// and s.qfree the free mask of the LAST index evaluation (1.0 free / 0.0 clamped).
// Returns 0 ok, TFMPC_ST_NOT_PD if the FIRST factorisation failed (the reference
// leaves Hfree unbound there), TFMPC_ST_QP_MAXITER after 100 iterations.
__device__ inline int boxqp(IlqrSmem &s, const float *H, int ldh, const float *q)
{
    const int m = s.m, lane = lane_id();
    const float rtol = 1e-8f, step_dec = 0.6f, min_step = 1e-22f, armijo = 0.1f, eps = 1e-6f;   // :13-17
    float value = qp_objective(H, ldh, q, s.qx, m);                                               // :22
    float old_value = value;
    for (int i = lane; i < m; i += kWave) s.qfree[i] = 1.0f;
    wsync();
    int rc = TFMPC_ST_QP_MAXITER;
    for (int it = 0; it < 100; ++it) {                                                            // :24
        if (it > 0 && (old_value - value) < rtol * fabsf(old_value)) { rc = 0; break; }           // :27-29
        old_value = value;
        // g = q + H x; clamped set (:34-35, :121-127); grad_clamped = q + H (x * clamped) (:65)
        int n_free_part = 0;
        float gn_part = 0.0f;
        for (int i = lane; i < m; i += kWave) {
            float g = q[i];
            for (int j = 0; j < m; ++j) g = fmaf(H[i * ldh + j], s.qx[j], g);
            const float x = s.qx[i];
            const bool clamped = (fabsf(x - s.qlo[i]) < eps && g > 0.0f) || (fabsf(s.qhi[i] - x) < eps && g < 0.0f);
            s.qg[i] = g;
            s.qfree[i] = clamped ? 0.0f : 1.0f;
            if (!clamped) { n_free_part += 1; gn_part = fmaf(g, g, gn_part); }
        }
        wsync();
        for (int i = lane; i < m; i += kWave) {
            float gc = q[i];
            for (int j = 0; j < m; ++j) gc = fmaf(H[i * ldh + j], s.qfree[j] != 0.0f ? 0.0f : s.qx[j], gc);
            s.qgc[i] = gc;
        }
        wsync();
        // factorise H_ff (Cholesky in the reference, :40-51) and solve for the Newton step
        const int bad = qp_solve_free(s, H, ldh, s.qfree, 1, [&](int i, int) { return s.qgc[i]; });
        if (bad) { rc = (it == 0) ? TFMPC_ST_NOT_PD : TFMPC_ST_QP_LATER_NOT_PD; break; }     // :47-51: the first failure raises, a later one breaks out
        const float n_free = wave_sum((float)n_free_part);
        if (n_free == 0.0f) { rc = 0; break; }                                                    // :53-55
        if (sqrtf(wave_sum(gn_part)) < eps) { rc = 0; break; }                                    // :58-62
        float sg_part = 0.0f;
        for (int i = lane; i < m; i += kWave) {                                                   // :66-72
            const float sr = s.qfree[i] != 0.0f ? (-s.aug[i * s.lda + m] - s.qx[i]) : 0.0f;
            s.qs[i] = sr;
            sg_part = fmaf(sr, s.qg[i], sg_part);
        }
        const float sdotg = wave_sum(sg_part);                                                    // :75
        if (sdotg >= 0.0f) { rc = 0; break; }                                                     // :77-79
        float step = 1.0f, vc;                                                                    // :82-95
        for (;;) {
            wsync();
            for (int i = lane; i < m; i += kWave)
                s.qc[i] = fminf(fmaxf(fmaf(step, s.qs[i], s.qx[i]), s.qlo[i]), s.qhi[i]);
            wsync();
            vc = qp_objective(H, ldh, q, s.qc, m);
            if (!((vc - old_value) / (step * sdotg) < armijo)) break;
            step *= step_dec;
            if (step < min_step) {      // the reference recomputes xc/vc once more before this test
                wsync();
                for (int i = lane; i < m; i += kWave)
                    s.qc[i] = fminf(fmaxf(fmaf(step, s.qs[i], s.qx[i]), s.qlo[i]), s.qhi[i]);
                wsync();
                vc = qp_objective(H, ldh, q, s.qc, m);
                break;
            }
        }
        for (int i = lane; i < m; i += kWave) s.qx[i] = s.qc[i];                                  // :98-99
        value = vc;
        wsync();
    }
    return rc;
}

// ---------------------------------------------------------- backward pass -----
struct BackwardResult {
    float J, dV1, dV2, g_norm;
    int failed;          // 1: Cholesky-type failure (tf.errors.InvalidArgumentError in the reference)
    int flags;           // TFMPC_ST_* accumulated (QP max-iter ...)
};

// Provider contract:
//   float load(int t)  -- fills s.fx, s.fu, s.lx, s.lu, s.lxx, s.luu, s.lux (= l_xu^T), s.uh (= u_hat_t)
//                         and returns l_t; called by all lanes; no trailing sync needed.
//   float load_final() -- fills s.Vx (= l_x^f) and s.Vxx (= l_xx^f), returns l^f.
// BLK: matrix products on the f32 matrix cores (wave_matmul_mfma: 16 x 16 tiles, one instruction per 1024
// multiply-adds).  Worth it from n ~ 12 up, where the lane-strided products are LDS-issue bound; it costs VGPRs,
// i.e. a wave per SIMD on small shapes, so the launchers pick the variant by shape (the same variant for every
// launch of a shape: the MFMA sums four products per accumulate, so the two are not bit-identical).
template <bool BLK, class FA, class FB, class FInit, class FOut>
__device__ __forceinline__ void matmul(int M, int N, int K, FA a, FB b, FInit init, FOut out)
{
    if constexpr (BLK) wave_matmul_mfma(M, N, K, a, b, init, out);
    else wave_matmul(M, N, K, a, b, init, out);
}

template <bool BLK = false, class Provider>
__device__ inline BackwardResult backward_pass(IlqrSmem &s, Provider &prov, int T, float mu, bool bounded,
                                               const float *low, const float *high, float *Kg, float *kg)
{
    const int n = s.n, m = s.m, ldn = s.ldn, ldm = s.ldm, lda = s.lda, lane = lane_id();
    BackwardResult r{0.0f, 0.0f, 0.0f, 0.0f, 0, 0};
    r.J = prov.load_final();                                           // ilqr.py:101-104
    float gsum = 0.0f;
    wsync();

    for (int t = T - 1; t >= 0; --t) {                                 // :108
        const float l = prov.load(t);
        wsync();
        // Q_x, Q_u (:122-123); W1 = f_x^T V_xx, W2 = f_u^T V_xx (:125-126)
        for (int i = lane; i < n + m; i += kWave) {
            float acc;
            // summation order: the diagonal term first, then the others by ascending row.  (The
            // register-resident costate kernel keeps column i of f_x in VGPRs with a zero on the
            // diagonal and adds the state-dependent diagonal entry separately; every path that forms
            // Q_x, Q_u uses this one order so that they stay bit-identical.)
            if (i < n) {
                acc = fmaf(s.fx[i * ldn + i], s.Vx[i], s.lx[i]);
                for (int kk = 0; kk < n; ++kk) if (kk != i) acc = fmaf(s.fx[kk * ldn + i], s.Vx[kk], acc);
                s.Qx[i] = acc;
            } else {
                const int a = i - n;
                acc = s.lu[a];
                if (a < n) acc = fmaf(s.fu[a * ldm + a], s.Vx[a], acc);
                for (int kk = 0; kk < n; ++kk) if (kk != a) acc = fmaf(s.fu[kk * ldm + a], s.Vx[kk], acc);
                s.Qu[a] = acc;
            }
        }
        matmul<BLK>(n, n, n, [&](int i, int kk) { return s.fx[kk * ldn + i]; }, [&](int kk, int j) { return s.Vxx[kk * ldn + j]; },
                    [](int, int) { return 0.0f; }, [&](int i, int j, float x) { s.W1[i * ldn + j] = x; });
        matmul<BLK>(m, n, n, [&](int a, int kk) { return s.fu[kk * ldm + a]; }, [&](int kk, int j) { return s.Vxx[kk * ldn + j]; },
                    [](int, int) { return 0.0f; }, [&](int a, int j, float x) { s.W2[a * ldn + j] = x; });
        // V_xx != 0 test of :137 (before V_xx is overwritten)
        float nzp = 0.0f;
        for (int idx = lane; idx < n * n; idx += kWave) nzp += (s.Vxx[(idx / n) * ldn + idx % n] != 0.0f) ? 1.0f : 0.0f;
        const bool vxx_nonzero = wave_sum(nzp) > 0.0f;
        wsync();
        // Q_xx, Q_uu, Q_ux (:129-131) and the regularised Q_uu, Q_ux (:127,133-134)
        matmul<BLK>(n, n, n, [&](int i, int kk) { return s.W1[i * ldn + kk]; }, [&](int kk, int j) { return s.fx[kk * ldn + j]; },
                    [&](int i, int j) { return s.lxx[i * ldn + j]; }, [&](int i, int j, float x) { s.Qxx[i * ldn + j] = x; });
        matmul<BLK>(m, m, n, [&](int a, int kk) { return s.W2[a * ldn + kk]; }, [&](int kk, int b) { return s.fu[kk * ldm + b]; },
                    [&](int a, int b) { return s.luu[a * ldm + b]; }, [&](int a, int b, float x) { s.Quu[a * ldm + b] = x; });
        matmul<BLK>(m, n, n, [&](int a, int kk) { return s.W2[a * ldn + kk]; }, [&](int kk, int j) { return s.fx[kk * ldn + j]; },
                    [&](int a, int j) { return s.lux[a * ldn + j]; }, [&](int a, int j, float x) { s.Qux[a * ldn + j] = x; });
        matmul<BLK>(m, m, n, [&](int a, int kk) { return fmaf(mu, s.fu[kk * ldm + a], s.W2[a * ldn + kk]); },
                    [&](int kk, int b) { return s.fu[kk * ldm + b]; },
                    [&](int a, int b) { return s.luu[a * ldm + b]; }, [&](int a, int b, float x) { s.Quur[a * ldm + b] = x; });
        matmul<BLK>(m, n, n, [&](int a, int kk) { return fmaf(mu, s.fu[kk * ldm + a], s.W2[a * ldn + kk]); },
                    [&](int kk, int j) { return s.fx[kk * ldn + j]; },
                    [&](int a, int j) { return s.lux[a * ldn + j]; }, [&](int a, int j, float x) { s.Quxr[a * ldn + j] = x; });
        wsync();

        if (!bounded) {
            // [k | K] = -Q~uu^-1 [Q_u | Q~ux]                          (:357-362)
            for (int idx = lane; idx < m * s.width; idx += kWave) {
                const int a = idx / s.width, j = idx - a * s.width;
                s.aug[a * lda + j] = j < m ? s.Quur[a * ldm + j] : (j == m ? s.Qu[a] : s.Quxr[a * ldn + (j - m - 1)]);
            }
            wsync();
            if (wave_gauss_jordan<false>(s.aug, lda, m, s.width, s.fac, s.prow)) { r.failed = 1; return r; }
            for (int idx = lane; idx < m * n; idx += kWave) s.K[(idx / n) * ldn + idx % n] = -s.aug[(idx / n) * lda + m + 1 + idx % n];
            for (int a = lane; a < m; a += kWave) s.k[a] = -s.aug[a * lda + m];
        } else if (vxx_nonzero) {
            // box-QP on (Q~uu, Q_u) with bounds low-u, high-u from the box centre   (:364-371)
            for (int a = lane; a < m; a += kWave) {
                const float lo = low[a] - s.uh[a], hi = high[a] - s.uh[a];
                s.qlo[a] = lo; s.qhi[a] = hi; s.qx[a] = (lo + hi) / 2;
            }
            wsync();
            const int rc = boxqp(s, s.Quur, ldm, s.Qu);
            // a LATER factorisation failure inside the QP (optimization.py:47-51 breaks out; ilqr.py:375-385 would then pair the stale factor
            // with the new free mask) fails the sweep like the first: the final free set is not positive definite, so K cannot be built
            // from it.  Unreachable in exact arithmetic: the QP starts at the box centre, all coordinates free (tfmpc_hip.h).
            if (rc == TFMPC_ST_NOT_PD || rc == TFMPC_ST_QP_LATER_NOT_PD) { r.failed = 1; return r; }
            r.flags |= rc;
            wsync();
            // K_free = -H_ff^-1 Q~ux[free], clamped rows 0                          (:375-385)
            if (qp_solve_free(s, s.Quur, ldm, s.qfree, n, [&](int a, int j) { return s.Quxr[a * ldn + j]; })) { r.failed = 1; return r; }
            for (int idx = lane; idx < m * n; idx += kWave) {
                const int a = idx / n, j = idx - a * n;
                s.K[a * ldn + j] = s.qfree[a] != 0.0f ? -s.aug[a * lda + m + j] : 0.0f;
            }
            for (int a = lane; a < m; a += kWave) s.k[a] = s.qx[a];
        } else {
            // V_xx == 0: K = 0, bang-bang k                                           (:140-141)
            for (int idx = lane; idx < m * n; idx += kWave) s.K[(idx / n) * ldn + idx % n] = 0.0f;
            for (int a = lane; a < m; a += kWave) s.k[a] = (s.Qu[a] >= 0.0f) ? (low[a] - s.uh[a]) : (high[a] - s.uh[a]);
        }
        wsync();

        // K^T Q_uu (:147)
        matmul<BLK>(n, m, m, [&](int i, int a) { return s.K[a * ldn + i]; }, [&](int a, int b) { return s.Quu[a * ldm + b]; },
                    [](int, int) { return 0.0f; }, [&](int i, int b, float x) { s.KtQ[i * ldm + b] = x; });
        wsync();
        // V_x (:149-154), V_xx before symmetrisation (:156-161) -> W1
        for (int i = lane; i < n; i += kWave) {
            float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            for (int a = 0; a < m; ++a) {
                s1 = fmaf(s.Qux[a * ldn + i], s.k[a], s1);
                s2 = fmaf(s.K[a * ldn + i], s.Qu[a], s2);
                s3 = fmaf(s.KtQ[i * ldm + a], s.k[a], s3);
            }
            s.Vx[i] = s.Qx[i] + s1 + s2 + s3;
        }
        for (int idx = lane; idx < n * n; idx += kWave) {
            const int i = idx / n, j = idx - i * n;
            float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            for (int a = 0; a < m; ++a) {
                const float Kaj = s.K[a * ldn + j];
                s1 = fmaf(s.Qux[a * ldn + i], Kaj, s1);
                s2 = fmaf(s.K[a * ldn + i], s.Qux[a * ldn + j], s2);
                s3 = fmaf(s.KtQ[i * ldm + a], Kaj, s3);
            }
            s.W1[i * ldn + j] = s.Qxx[i * ldn + j] + s1 + s2 + s3;
        }
        // J, dV1, dV2 (:164-167) and the g_norm term of solve (:243)
        float p1 = 0.0f, p2 = 0.0f, gmax = 0.0f;
        for (int a = lane; a < m; a += kWave) {
            float quk = 0.0f;
            for (int b = 0; b < m; ++b) quk = fmaf(s.Quu[a * ldm + b], s.k[b], quk);
            p1 = fmaf(s.k[a], s.Qu[a], p1);
            p2 = fmaf(s.k[a], quk, p2);
            gmax = fmaxf(gmax, fabsf(s.k[a]) / (fabsf(s.uh[a]) + 1.0f));
        }
        r.J += l;
        r.dV1 += wave_sum(p1);
        r.dV2 += 0.5f * wave_sum(p2);
        gsum += wave_max(gmax);
        // gains out
        for (int idx = lane; idx < m * n; idx += kWave) Kg[(size_t)t * m * n + idx] = stq(s, s.K[(idx / n) * ldn + idx % n]);
        for (int a = lane; a < m; a += kWave) kg[(size_t)t * m + a] = stq(s, s.k[a]);
        wsync();
        for (int idx = lane; idx < n * n; idx += kWave) {                                 // :162
            const int i = idx / n, j = idx - i * n;
            s.Vxx[i * ldn + j] = 0.5f * (s.W1[i * ldn + j] + s.W1[j * ldn + i]);
        }
        wsync();
    }
    r.g_norm = T > 0 ? gsum / (float)T : 0.0f;
    return r;
}

// Backward pass of ilqr.py:94-172 when l_xx = l_uu = l_ux = 0 and l^f_xx = 0 and the action
// box is bounded: V_xx stays exactly 0, the controller is always the bang-bang branch
// (ilqr.py:140-141), K_t = 0, and what is left is the adjoint (costate) recursion
//   Q_x = l_x + f_x^T V_x,  Q_u = l_u + f_u^T V_x,  k = Q_u >= 0 ? low - u : high - u,  V_x <- Q_x.
// Bit-identical to backward_pass on such envs (every dropped term is an exact zero).
template <int KIND>
__device__ inline BackwardResult backward_pass_adjoint(IlqrSmem &s, const EnvLds &e, int T, const float *xhat,
                                                       const float *uhat, float *kg)
{
    const int n = s.n, m = s.m, lane = lane_id();
    BackwardResult r{0.0f, 0.0f, 0.0f, 0.0f, 0, 0};
    for (int i = lane; i < n; i += kWave) s.xh[i] = xhat[(size_t)T * n + i];
    wsync();
    r.J = Env<KIND>::final_grad(e, s.xh, s.Vx);
    float gsum = 0.0f;
    wsync();
    // The nominal trajectory streams in from the HBM workspace one step ahead of its use (each lane
    // keeps the next step's element in a register), and f_x, f_u are never materialised: the env
    // forms the entries it needs on the fly (Env<KIND>::adjoint_qx / adjoint_qu), in the summation
    // order of the dense path.
    const bool one_pass = n + m <= kWave;           // lane i < n: state i; lane n + a: action a
    float x_next = 0.0f, u_next = 0.0f;
    if (one_pass && T > 0) {
        if (lane < n) x_next = xhat[(size_t)(T - 1) * n + lane];
        else if (lane < n + m) u_next = uhat[(size_t)(T - 1) * m + (lane - n)];
    }
    for (int t = T - 1; t >= 0; --t) {
        if (one_pass) {
            if (lane < n) s.xh[lane] = x_next;
            else if (lane < n + m) s.uh[lane - n] = u_next;
            if (t > 0) {
                if (lane < n) x_next = xhat[(size_t)(t - 1) * n + lane];
                else if (lane < n + m) u_next = uhat[(size_t)(t - 1) * m + (lane - n)];
            }
        } else {
            for (int i = lane; i < n; i += kWave) s.xh[i] = xhat[(size_t)t * n + i];
            for (int a = lane; a < m; a += kWave) s.uh[a] = uhat[(size_t)t * m + a];
        }
        wsync();
        float l = 0.0f;
        float p1 = 0.0f, gmax = 0.0f;
        if constexpr (KIND == TFMPC_ENV_USER) {
            // a user env (user_env.h): direction i of z = [x; u] in lane i, Q_z[i] by ONE first-order dual evaluation of the user's transition and cost
            for (int base = 0; base < n + m; base += kWave) {
                const int i = base + lane;
                float acc;
                l = Env<KIND>::adjoint_direction(e, s.xh, s.uh, s.Vx, i, acc);
                if (i < n) {
                    s.Qx[i] = acc;
                } else if (i < n + m) {
                    const int a = i - n;
                    const float kt = (acc >= 0.0f) ? (e.low[a] - s.uh[a]) : (e.high[a] - s.uh[a]);       // ilqr.py:140-141
                    kg[(size_t)t * m + a] = stq(s, kt);
                    p1 = fmaf(kt, acc, p1);
                    gmax = fmaxf(gmax, fabsf(kt) / (fabsf(s.uh[a]) + 1.0f));
                }
            }
        } else {
        l = Env<KIND>::cost(e, s.xh, s.uh);
        for (int i = lane; i < n + m; i += kWave) {
            if (i < n) {
                s.Qx[i] = Env<KIND>::adjoint_qx(e, s.xh, s.uh, s.Vx, Env<KIND>::cost_grad_x_i(e, s.xh, i), i);
            } else {
                const int a = i - n;
                const float acc = Env<KIND>::adjoint_qu(e, s.xh, s.uh, s.Vx, a);
                const float kt = (acc >= 0.0f) ? (e.low[a] - s.uh[a]) : (e.high[a] - s.uh[a]);
                kg[(size_t)t * m + a] = stq(s, kt);
                p1 = fmaf(kt, acc, p1);
                gmax = fmaxf(gmax, fabsf(kt) / (fabsf(s.uh[a]) + 1.0f));
            }
        }
        }
        r.J += l;
        r.dV1 += wave_sum(p1);
        gsum += wave_max(gmax);
        wsync();
        for (int i = lane; i < n; i += kWave) s.Vx[i] = s.Qx[i];
        wsync();
    }
    r.g_norm = T > 0 ? gsum / (float)T : 0.0f;
    return r;
}

// ------------------------------------------------------------ rollouts --------
// Open-loop rollout under given actions (iLQR.start, ilqr.py:53-82).
template <int KIND>
__device__ inline void rollout_pass(IlqrSmem &s, const EnvLds &e, int T, const float *x0, const float *actions,
                                    float *states, float *costs, float *actions_out)
{
    const int n = s.n, m = s.m, lane = lane_id();
    for (int i = lane; i < n; i += kWave) { const float x = x0[i]; s.xv[i] = x; states[i] = stq(s, x); }
    for (int t = 0; t < T; ++t) {
        for (int a = lane; a < m; a += kWave) {
            const float u = actions[(size_t)t * m + a];
            s.uv[a] = u;
            if (actions_out) actions_out[(size_t)t * m + a] = stq(s, u);
        }
        wsync();
        const float c = Env<KIND>::cost(e, s.xv, s.uv);
        Env<KIND>::transition(e, s.xv, s.uv, s.xn);
        if (lane == 0) costs[t] = c;
        wsync();
        for (int i = lane; i < n; i += kWave) { const float x = s.xn[i]; s.xv[i] = x; states[(size_t)(t + 1) * n + i] = stq(s, x); }
    }
    wsync();
    const float fc = Env<KIND>::final_cost(e, s.xv);
    if (lane == 0) costs[T] = fc;
}

// Closed-loop rollout with step alpha (iLQR.forward, ilqr.py:174-212).
template <int KIND, bool HAS_K = true>
__device__ inline void forward_pass(IlqrSmem &s, const EnvLds &e, int T, float alpha, const float *xhat,
                                    const float *uhat, const float *Kg, const float *kg, float *states,
                                    float *actions, float *costs, float &J_out, float &residual_out)
{
    const int n = s.n, m = s.m, ldn = s.ldn, lane = lane_id();
    for (int i = lane; i < n; i += kWave) { const float x = xhat[i]; s.xv[i] = x; states[i] = stq(s, x); }
    float J = 0.0f, resid = 0.0f;
    if constexpr (!HAS_K) {
        if (n <= kWave && m <= kWave) {
            // K == 0 (adjoint envs): u_t = clip(u_hat_t + alpha k_t) needs no state feedback, so the
            // step streams (k_t, u_hat_t) one step ahead in registers, keeps the running max |du| per
            // lane (one reduction at the end) and ping-pongs the state between two LDS vectors.
            float *xcur = s.xv, *xnext = s.xn;
            float k_n = 0.0f, uh_n = 0.0f, rmax = 0.0f;
            if (lane < m && T > 0) { k_n = kg[lane]; uh_n = uhat[lane]; }
            for (int t = 0; t < T; ++t) {
                const float k_c = k_n, uh_c = uh_n;
                if (lane < m && t + 1 < T) { k_n = kg[(size_t)(t + 1) * m + lane]; uh_n = uhat[(size_t)(t + 1) * m + lane]; }
                if (lane < m) {
                    const float du = alpha * k_c;                                            // :193-194
                    const float u = fminf(fmaxf(uh_c + du, e.low[lane]), e.high[lane]);      // :196-197
                    s.uv[lane] = u;
                    actions[(size_t)t * m + lane] = stq(s, u);
                    rmax = fmaxf(rmax, fabsf(du));
                }
                wsync();
                const float c = Env<KIND>::cost(e, xcur, s.uv);                              // :198
                Env<KIND>::transition(e, xcur, s.uv, xnext);                                 // :199
                J += c;                                                                      // :205
                if (lane == 0) costs[t] = c;
                wsync();
                if (lane < n) states[(size_t)(t + 1) * n + lane] = stq(s, xnext[lane]);
                float *tmp = xcur; xcur = xnext; xnext = tmp;
            }
            wsync();
            const float fc = Env<KIND>::final_cost(e, xcur);                                 // :208-210
            if (lane == 0) costs[T] = fc;
            J_out = J + fc;
            residual_out = wave_max(rmax);                                                   // :206
            return;
        }
    }
    for (int t = 0; t < T; ++t) {
        if (HAS_K) load_matrix(s.K, ldn, Kg + (size_t)t * m * n, m, n);
        for (int a = lane; a < m; a += kWave) { s.k[a] = kg[(size_t)t * m + a]; s.uh[a] = uhat[(size_t)t * m + a]; }
        for (int i = lane; i < n; i += kWave) s.xh[i] = xhat[(size_t)t * n + i];
        wsync();
        float rmax = 0.0f;
        for (int a = lane; a < m; a += kWave) {
            float du = alpha * s.k[a];                                                   // :193-194
            if (HAS_K) for (int j = 0; j < n; ++j) du = fmaf(s.K[a * ldn + j], s.xv[j] - s.xh[j], du);
            const float u = fminf(fmaxf(s.uh[a] + du, e.low[a]), e.high[a]);             // :196-197
            s.uv[a] = u;
            actions[(size_t)t * m + a] = stq(s, u);
            rmax = fmaxf(rmax, fabsf(du));
        }
        resid = fmaxf(resid, wave_max(rmax));                                            // :206
        wsync();
        const float c = Env<KIND>::cost(e, s.xv, s.uv);                                  // :198
        Env<KIND>::transition(e, s.xv, s.uv, s.xn);                                      // :199
        J += c;                                                                          // :205
        if (lane == 0) costs[t] = c;
        wsync();
        for (int i = lane; i < n; i += kWave) { const float x = s.xn[i]; s.xv[i] = x; states[(size_t)(t + 1) * n + i] = stq(s, x); }
    }
    wsync();
    const float fc = Env<KIND>::final_cost(e, s.xv);                                     // :208-210
    if (lane == 0) costs[T] = fc;
    J_out = J + fc;
    residual_out = resid;
}

}  // namespace tfmpc
