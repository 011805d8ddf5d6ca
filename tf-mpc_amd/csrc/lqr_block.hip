// lqr_block.hip -- batched LQR for LARGE shapes on gfx950: one workgroup of four wavefronts per problem
// instance, every operand of the instance in that workgroup's LDS, the dense products on the fp32 matrix
// cores (v_mfma_f32_16x16x4_f32: fp32 in, fp32 accumulate -- no precision trade).
//
// Same recursion as lqr_generic.hip (tfmpc/solvers/lqr.py: LQR.backward :59-129, LQR.forward :131-161,
// LQR.solve :163-166) and the same outputs.  It exists because a single wave per instance stops scaling around
// n + m ~ 24: its LDS slice then allows one wave per SIMD, and that wave is latency-bound on its own instruction
// stream (measured at n = 32, m = 16: 117 k cycles per Riccati step, 37 % of them in the elimination).  Four
// waves split every phase and three workgroups per CU overlap each other's LDS round trips.
//
// The vectors ride along as an extra column of their matrix -- [F | f], [C | c], [V | v], [K | k] -- so that
// F^T v, W f, Q_xu k, K^T q_u and K^T Q_uu k come out of the matrix products instead of being lane-serial
// mat-vecs:   [W | F^T v] = F^T [V | v]                                   (lqr.py:74,78)
//             [Q | c + W f] = [C | c] + W [F | f] ;  q = that column + F^T v  (lqr.py:75-78)
//             [V' | v'] = [Q_xx | q_x] + [Q_xu | K^T | K^T Q_uu] [[K | k]; [Q_ux | q_u]; [K | k]]   (lqr.py:97-105)
#include <hip/hip_runtime.h>

#include "block_ops.h"
#include "lqr_kernels.h"

namespace tfmpc {

namespace {

constexpr int kBW = 4;                       // waves per instance

// -DTFMPC_PHASE_PROBE: instance 0 accumulates s_memtime deltas per phase of the Riccati step and leaves them in
// K[0][0][:8] (tools/probes/block_probe.py); compiled out otherwise.
#ifdef TFMPC_PHASE_PROBE
#define PHASE(i) { const long long now = __builtin_readcyclecounter(); pc[i] += now - tk; tk = now; }
#else
#define PHASE(i)
#endif
constexpr int kBT = kBW * kWave;

struct BlockSmem {
    int ldf, ldv, ldm, lda, width;
    int scratch_floats;
    float *Fa, *Ca, *Vv[2], *Wv, *Qq, *q, *aug, *fac, *prow, *rowp, *Kk, *KtQ, *z, *y, *red, *part, *scratch;
};

__host__ __device__ inline size_t block_smem_floats(int n, int m)
{
    const int d = n + m, ldf = odd_ld(d + 1), ldv = odd_ld(n + 1), ldm = odd_ld(m), width = m + 1 + n, lda = odd_ld(width);
    size_t s = 0;
    s += (size_t)n * ldf;            // [F | f]
    s += (size_t)d * ldf;            // [C | c]
    s += 2 * (size_t)n * ldv;        // [V | v], double-buffered
    s += (size_t)d * ldv;            // [W | F^T v]
    s += (size_t)d * ldf + d;        // [Q | c + W f], q
    s += (size_t)m * lda + m + 2 * width;   // aug, fac, prow, rowp
    s += (size_t)m * ldv;            // [K | k]
    s += (size_t)n * ldm;            // K^T Q_uu
    s += (size_t)d + d + n;          // z, y = [C z ; F z]
    s += kBW + kBT;                  // reduction scratch
    return s;
}

__device__ inline BlockSmem block_carve(float *base, int n, int m)
{
    BlockSmem s;
    const int d = n + m;
    s.ldf = odd_ld(d + 1);
    s.ldv = odd_ld(n + 1);
    s.ldm = odd_ld(m);
    s.width = m + 1 + n;
    s.lda = odd_ld(s.width);
    float *p = base;
    s.Fa = p; p += n * s.ldf;
    s.Ca = p; p += d * s.ldf;
    s.Kk = p; p += m * s.ldv;
    s.z = p; p += d;
    s.y = p; p += d + n;
    s.red = p; p += kBW;
    s.part = p; p += kBT;
    // from here on: operands of the Riccati sweep only -- the rollout reuses the region for its cost post-pass
    s.scratch = p;
    s.Vv[0] = p; p += n * s.ldv;
    s.Vv[1] = p; p += n * s.ldv;
    s.Wv = p; p += d * s.ldv;
    s.Qq = p; p += d * s.ldf;
    s.q = p; p += d;
    s.aug = p; p += m * s.lda;
    s.fac = p; p += m;
    s.prow = p; p += s.width;
    s.rowp = p; p += s.width;
    s.KtQ = p; p += n * s.ldm;
    s.scratch_floats = (int)(p - s.scratch);
    return s;
}

template <bool BACKWARD, bool FORWARD>
__global__ __launch_bounds__(kBT) __attribute__((amdgpu_waves_per_eu(3, 3))) void lqr_block_kernel(LqrArgs a)
{
    extern __shared__ float smem[];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = a.n, m = a.m, d = n + m, T = a.T;
    BlockSmem s = block_carve(smem, n, m);
    const int ldf = s.ldf, ldv = s.ldv, ldm = s.ldm, lda = s.lda;

    {
        const float *Fg = a.F + (size_t)b * a.sF, *Cg = a.C + (size_t)b * a.sC;
        block_for_2d<kBW>(n, d, [&](int i, int j, int idx) { s.Fa[i * ldf + j] = Fg[idx]; });
        block_for_2d<kBW>(d, d, [&](int i, int j, int idx) { s.Ca[i * ldf + j] = Cg[idx]; });
        for (int i = tid; i < n; i += kBT) s.Fa[i * ldf + d] = a.f[(size_t)b * a.sf + i];
        for (int i = tid; i < d; i += kBT) s.Ca[i * ldf + d] = a.c[(size_t)b * a.sc + i];
    }
    __syncthreads();

    int status = 0;
    float *Kg = a.K ? a.K + (size_t)b * a.sK : nullptr;
    float *kg = a.k ? a.k + (size_t)b * a.sk : nullptr;

    if (BACKWARD) {
        // terminal condition V = C_xx, v = c_x, const = 0              (lqr.py:67-69)
        block_for_2d<kBW>(n, n + 1, [&](int i, int j, int) { s.Vv[0][i * ldv + j] = s.Ca[i * ldf + (j < n ? j : d)]; });
        float cst = 0.0f;
        int cur = 0;
        __syncthreads();

#ifdef TFMPC_PHASE_PROBE
        long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk = __builtin_readcyclecounter();
#endif
        for (int t = T - 1; t >= 0; --t) {
            const int v_size = n * ldv;                         // [V | v] buffers sit back to back: pick by offset
            const float *Vc = s.Vv[0] + (cur ? v_size : 0);     // (keeps the pointers provably in LDS)
            float *Vn = s.Vv[0] + (cur ? 0 : v_size);
            // [W | F^T v] = F^T [V | v]   [d][n+1]                      (lqr.py:74,78)
            block_matmul_mfma<kBW>(d, n + 1, n,
                        [&](int r, int k) { return s.Fa[k * ldf + r]; },
                        [&](int k, int j) { return Vc[k * ldv + j]; },
                        [](int, int) { return 0.0f; },
                        [&](int r, int j, float x) { s.Wv[r * ldv + j] = x; });
            __syncthreads();
            PHASE(0)
            // [Q | c + W f] = [C | c] + W [F | f]   [d][d+1]            (lqr.py:75-78)
            block_matmul_mfma<kBW>(d, d + 1, n,
                        [&](int r, int k) { return s.Wv[r * ldv + k]; },
                        [&](int k, int j) { return s.Fa[k * ldf + j]; },
                        [&](int r, int j) { return s.Ca[r * ldf + j]; },
                        [&](int r, int j, float x) { s.Qq[r * ldf + j] = x; });
            __syncthreads();
            PHASE(1)
            // q = c + W f + F^T v;  [K | k] = -Q_uu^-1 [Q_ux | q_u]                    (lqr.py:84-87)
            for (int r = tid; r < d; r += kBT) s.q[r] = s.Qq[r * ldf + d] + s.Wv[r * ldv + n];
            auto system = [&](int r, int j) {            // [Q_uu | q_u | Q_ux], straight from the product tiles:
                const int col = j < m ? n + j : (j == m ? d : j - m - 1);       // one read of row n + r of [Q | .] ...
                const float x = s.Qq[(n + r) * ldf + col];
                const float fv = s.Wv[(n + r) * ldv + n];                       // ... plus (F^T v) in the q_u column
                return j == m ? x + fv : x;
            };
            if (m <= 16 && s.width <= kWave) {
                // one wave eliminates in registers (no pivoting: Q_uu of a convex problem is positive definite,
                // and a pivot <= 0 is reported, like the matrix-core kernel does); the other waves go to the barrier
                if (Block<kBW>::wave() == 0) {
                    float col[16];
                    const int bad = wave_gj16_registers(m, s.width, system, col);
                    if (bad) status |= (bad & 2) ? TFMPC_ST_NOT_PD : TFMPC_ST_SINGULAR;
                    const int j = Block<kBW>::lane() - m - 1;            // -1: the k column, 0..n-1: columns of K
                    if (j >= -1 && j < n) {
                        float *lds_dst = s.Kk + (j < 0 ? n : j);
                        float *dst = j < 0 ? kg + (size_t)t * m : Kg + (size_t)t * m * n + j;
                        const int stride = j < 0 ? 1 : n;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (r < m) {
                                const float x = -col[r];
                                lds_dst[r * ldv] = x;
                                dst[r * stride] = x;
                            }
                        }
                    }
                }
            } else {
                // general sizes: the augmented system in LDS, all four waves, row pivoting (the general inverse)
                block_for_2d<kBW>(m, s.width, [&](int r, int j, int) { s.aug[r * lda + j] = system(r, j); });
                __syncthreads();
                if (block_gauss_jordan<true, kBW>(s.aug, lda, m, s.width, s.fac, s.prow, s.rowp)) status |= TFMPC_ST_SINGULAR;
                block_for_2d<kBW>(m, n, [&](int r, int j, int idx) {
                    const float x = -s.aug[r * lda + m + 1 + j];
                    s.Kk[r * ldv + j] = x;
                    if (Kg) Kg[(size_t)t * m * n + idx] = x;
                });
                for (int r = tid; r < m; r += kBT) {
                    const float x = -s.aug[r * lda + m];
                    s.Kk[r * ldv + n] = x;
                    if (kg) kg[(size_t)t * m + r] = x;
                }
            }
            PHASE(3)
            __syncthreads();
            PHASE(4)
            // K^T Q_uu  [n][m]                                         (lqr.py:95)
            block_matmul_mfma<kBW>(n, m, m,
                        [&](int i, int k) { return s.Kk[k * ldv + i]; },
                        [&](int k, int j) { return s.Qq[(n + k) * ldf + n + j]; },
                        [](int, int) { return 0.0f; },
                        [&](int i, int j, float x) { s.KtQ[i * ldm + j] = x; });
            // const += 1/2 k^T Q_uu k + k^T q_u + 1/2 f^T V f + f^T v  (lqr.py:113-121, with the value
            // function BEFORE this step's update), only when it is an output: rows of Q_uu k on the first
            // threads, rows of V f after them
            if (a.cst) {
                float part = 0.0f;
                for (int r = tid; r < m + n; r += kBT) {
                    if (r < m) {
                        float quk = 0.0f;
                        for (int k = 0; k < m; ++k) quk = fmaf(s.Qq[(n + r) * ldf + n + k], s.Kk[k * ldv + n], quk);
                        part += s.Kk[r * ldv + n] * (0.5f * quk + s.q[n + r]);
                    } else {
                        const int i = r - m;
                        float vf = 0.0f;
                        for (int k = 0; k < n; ++k) vf = fmaf(Vc[i * ldv + k], s.Fa[k * ldf + d], vf);
                        part += s.Fa[i * ldf + d] * (0.5f * vf + Vc[i * ldv + n]);
                    }
                }
                cst += block_sum<kBW>(part, s.red);
            }
            __syncthreads();                             // K^T Q_uu is complete
            PHASE(5)
            // [V' | v'] = [Q_xx | q_x] + [Q_xu | K^T | K^T Q_uu] [[K | k]; [Q_ux | q_u]; [K | k]]   (lqr.py:97-105)
            block_matmul_mfma<kBW>(n, n + 1, 3 * m,
                        [&](int i, int k) {                  // one read from a per-lane address (k differs across lanes)
                            const float *p0 = s.Qq + i * ldf + n + k, *p1 = s.Kk + (k - m) * ldv + i, *p2 = s.KtQ + i * ldm + (k - 2 * m);
                            return *(k < m ? p0 : (k < 2 * m ? p1 : p2));
                        },
                        [&](int k, int j) {
                            const int kk = k < m ? k : (k < 2 * m ? k - m : k - 2 * m);
                            const float *pk = s.Kk + kk * ldv + j;
                            const float *pq = j < n ? s.Qq + (n + kk) * ldf + j : s.q + n + kk;
                            return *((k >= m && k < 2 * m) ? pq : pk);
                        },
                        [&](int i, int j) { return *(j < n ? s.Qq + i * ldf + j : s.q + i); },
                        [&](int i, int j, float x) { Vn[i * ldv + j] = x; });
            __syncthreads();
            PHASE(6)
            if (a.V) block_for_2d<kBW>(n, n, [&](int i, int j, int idx) { a.V[((size_t)b * T + t) * n * n + idx] = Vn[i * ldv + j]; });
            if (a.v) for (int i = tid; i < n; i += kBT) a.v[((size_t)b * T + t) * n + i] = Vn[i * ldv + n];
            if (a.cst && tid == 0) a.cst[(size_t)b * T + t] = cst;
            cur ^= 1;
            PHASE(7)
        }
#ifdef TFMPC_PHASE_PROBE
        if (b == 0 && tid == 0 && Kg) for (int i = 0; i < 8; ++i) Kg[i] = (float)pc[i];
#endif
        // a NaN anywhere in the recursion ends up in the value function
        int nan = 0;
        block_for_2d<kBW>(n, n + 1, [&](int i, int j, int) { const float x = (s.Vv[0] + (cur ? n * ldv : 0))[i * ldv + j]; nan |= !(x == x); });
        if (__syncthreads_or(nan) || !(cst == cst)) status |= TFMPC_ST_NAN;
    }

    if (FORWARD) {
        const float *Kr = a.K + (size_t)b * a.sK;
        const float *kr = a.k + (size_t)b * a.sk;
        float *xs = a.states + (size_t)b * (T + 1) * n;
        float *us = a.actions + (size_t)b * T * m;
        float *cs = a.costs + (size_t)b * (T + 1);
        __syncthreads();
        for (int i = tid; i < n; i += kBT) {
            const float x = a.x0[(size_t)b * n + i];
            s.z[i] = x;
            xs[i] = x;
        }
        // The rollout proper is u = K x + k, x' = F z + f per step; the stage costs 1/2 z^T C z + c^T z (lqr.py:41-47)
        // need no sequencing, so the visited z_t are kept as columns of Z (in the sweep's dead LDS) and priced
        // afterwards, a chunk of timesteps at a time, as Y = C Z on the matrix cores.  The final cost
        // 1/2 x^T C_xx x + c_x^T x (lqr.py:49-57) is the same expression at z = [x_T; 0].
        const int cap = s.scratch_floats / (2 * d);           // columns of Z and of Y that fit (>= 1: the region holds [Q | q])
        const int ldz = (cap & 1) ? cap : cap - 1;            // odd leading dimension
        const int tc = T + 1 < ldz ? T + 1 : ldz;             // timesteps per chunk
        float *Z = s.scratch, *Y = s.scratch + d * ldz;
        constexpr int kPre = 8;                               // gains of the next step, prefetched into registers
        const int gains = m * (n + 1);                        // K_t then k_t, as they lie in HBM
        const bool prefetch = gains <= kPre * kBT;
        float gain[kPre];
        int gain_at[kPre];                                    // where each lands in [K | k]
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int idx = tid + u * kBT;
            gain_at[u] = idx < m * n ? (idx / n) * ldv + idx % n : (idx - m * n) * ldv + n;
        }
        auto fetch_gains = [&](int t) {
#pragma unroll
            for (int u = 0; u < kPre; ++u) {
                const int idx = tid + u * kBT;
                gain[u] = idx < m * n ? Kr[(size_t)t * m * n + idx] : (idx < gains ? kr[(size_t)t * m + idx - m * n] : 0.0f);
            }
        };
        if (prefetch && T > 0) fetch_gains(0);
        int nan = 0;
        for (int t0 = 0; t0 <= T; t0 += tc) {
            const int cols = (T + 1 - t0 < tc) ? T + 1 - t0 : tc;          // timesteps t0 .. t0 + cols - 1 (T = final state)
            for (int tt = 0; tt < cols; ++tt) {
                const int t = t0 + tt;
                if (t == T) {
                    for (int r = tid; r < d; r += kBT) Z[r * ldz + tt] = r < n ? s.z[r] : 0.0f;
                    break;
                }
                if (prefetch) {
#pragma unroll
                    for (int u = 0; u < kPre; ++u)
                        if (tid + u * kBT < gains) s.Kk[gain_at[u]] = gain[u];
                    if (t + 1 < T) fetch_gains(t + 1);
                } else {
                    block_for_2d<kBW>(m, n, [&](int r, int j, int idx) { s.Kk[r * ldv + j] = Kr[(size_t)t * m * n + idx]; });
                    for (int r = tid; r < m; r += kBT) s.Kk[r * ldv + n] = kr[(size_t)t * m + r];
                }
                __syncthreads();
                // u = K x + k                                                (lqr.py:143)
                block_matvec<kBW>(m, n,
                            [&](int r, int k) { return s.Kk[r * ldv + k]; },
                            [&](int k) { return s.z[k]; },
                            [&](int r) { return s.Kk[r * ldv + n]; }, s.part,
                            [&](int r, float u) { s.z[n + r] = u; us[(size_t)t * m + r] = u; });
                // x' = F z + f                                                 (lqr.py:36-39)
                block_matvec<kBW>(n, d,
                            [&](int r, int k) { return s.Fa[r * ldf + k]; },
                            [&](int k) { return s.z[k]; },
                            [&](int r) { return s.Fa[r * ldf + d]; }, s.part,
                            [&](int r, float v) { s.y[r] = v; });
                for (int r = tid; r < d; r += kBT) Z[r * ldz + tt] = s.z[r];
                __syncthreads();
                for (int i = tid; i < n; i += kBT) {
                    const float x = s.y[i];
                    s.z[i] = x;
                    xs[(size_t)(t + 1) * n + i] = x;
                }
                __syncthreads();
            }
            __syncthreads();
            block_matmul_mfma<kBW>(d, cols, d,
                        [&](int r, int k) { return s.Ca[r * ldf + k]; },
                        [&](int k, int j) { return Z[k * ldz + j]; },
                        [](int, int) { return 0.0f; },
                        [&](int r, int j, float v) { Y[r * ldz + j] = v; });
            __syncthreads();
            for (int j = tid; j < cols; j += kBT) {
                float cost = 0.0f;
                for (int r = 0; r < d; ++r) cost = fmaf(Z[r * ldz + j], fmaf(0.5f, Y[r * ldz + j], s.Ca[r * ldf + d]), cost);
                cs[t0 + j] = cost;
                nan |= !(cost == cost);
            }
            __syncthreads();
        }
        if (__syncthreads_or(nan)) status |= TFMPC_ST_NAN;
    }

    if (a.status && tid == 0) a.status[b] = status;
}

template <bool BW, bool FW>
int launch(const LqrArgs &a, hipStream_t stream)
{
    const size_t smem = lqr_block_smem_bytes(a.n, a.m);
    if (smem > kMaxLdsBytes) return TFMPC_ERR_UNSUPPORTED;
    auto kern = lqr_block_kernel<BW, FW>;
    if (smem > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return TFMPC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(kBT), smem, stream, a);
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}

}  // namespace

size_t lqr_block_smem_bytes(int n, int m) { return block_smem_floats(n, m) * sizeof(float); }

int lqr_block_launch(const LqrArgs &a, bool backward, bool forward, hipStream_t stream)
{
    if (backward && forward) return launch<true, true>(a, stream);
    if (backward) return launch<true, false>(a, stream);
    return launch<false, true>(a, stream);
}

}  // namespace tfmpc
