// wave_ops.h -- building blocks for "one wavefront owns one problem instance".
//
// gfx950 only: a wavefront is 64 lanes, a workgroup here is exactly one wave, and
// every matrix of the instance lives in that wave's LDS slice.  Because the
// workgroup is a single wave, __syncthreads() costs an s_waitcnt (the s_barrier
// is elided by the compiler) and is used purely as the LDS/global ordering fence
// between a phase that writes a tile and the phase that reads it across lanes.
#pragma once

#include <hip/hip_runtime.h>

namespace tfmpc {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// Fence between cross-lane producer/consumer phases of one wave.
__device__ __forceinline__ void wsync() { __syncthreads(); }

// The same ordering point for phases that talk through LDS ONLY (round 6).  __syncthreads() is a fence over EVERY address space: in a
// one-wave workgroup it compiles to s_waitcnt vmcnt(0) lgkmcnt(0), i.e. it also waits for every global load and store in flight -- inside a
// time loop that parks the wave on the round trip of the gains it has just PREFETCHED for the next step, or on the acknowledgement of the
// gains it has just stored, once per barrier.  A fence restricted to the local address space keeps the LDS order (s_waitcnt lgkmcnt(0);
// a wave's LDS instructions execute in issue order) and lets the compiler count the global accesses instead (s_waitcnt vmcnt(N) where a
// loaded value is used).  Global data written by one lane and read by another still needs wsync() between the two phases.
// -DTFMPC_FULL_SYNC restores __syncthreads() everywhere (A/B builds).
__device__ __forceinline__ void lds_sync()
{
#ifdef TFMPC_FULL_SYNC
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

// Full-wave reductions on the vector unit's DPP cross-lane paths (no LDS traffic: `__shfl_xor`
// compiles to ds_bpermute_b32 on gfx950, six dependent LDS-crossbar round trips per reduction).
// Four butterfly steps inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 fold the four
// rows into lane 63, which a v_readlane hands back as a wave-uniform value.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_move(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E, kDppRowHalfMirror = 0x141, kDppRowMirror = 0x140,
              kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_move<kDppQuadXor1>(0.0f, v);
    v += dpp_move<kDppQuadXor2>(0.0f, v);
    v += dpp_move<kDppRowHalfMirror>(0.0f, v);
    v += dpp_move<kDppRowMirror>(0.0f, v);                 // every lane: sum of its row of 16
    v += dpp_move<kDppRowBcast15, 0xA>(0.0f, v);           // rows 1, 3 += row 0, 2
    v += dpp_move<kDppRowBcast31, 0xC>(0.0f, v);           // rows 2, 3 += rows 0..1
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Maximum over the wave of NON-NEGATIVE, non-NaN values (every caller reduces |.| quantities built with
// fmaxf from 0): for those the IEEE order is the unsigned order of the bit patterns, and v_max_u32 takes
// its DPP operand directly -- one instruction per step instead of move + canonicalise + v_max_f32.
__device__ __forceinline__ float wave_max(float v)
{
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto step = [&](auto mv) { const unsigned o = (unsigned)mv; u = u > o ? u : o; };
    step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppQuadXor1, 0xF, 0xF, false));
    step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppQuadXor2, 0xF, 0xF, false));
    step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppRowHalfMirror, 0xF, 0xF, false));
    step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppRowMirror, 0xF, 0xF, false));
    step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppRowBcast15, 0xA, 0xF, false));
    step(__builtin_amdgcn_update_dpp((int)u, (int)u, kDppRowBcast31, 0xC, 0xF, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)u, 63));
}

// sum / exchange over groups of 2 or 4 adjacent lanes (a quad)
__device__ __forceinline__ float quad_xor1(float v) { return dpp_move<kDppQuadXor1>(0.0f, v); }
__device__ __forceinline__ float quad_xor2(float v) { return dpp_move<kDppQuadXor2>(0.0f, v); }

// Odd leading dimension: a column walk (stride ld) then touches every LDS bank.
__host__ __device__ __forceinline__ int odd_ld(int x) { return x | 1; }

// out(i, j) = init(i, j) + sum_k a(i, k) * b(k, j), i < M, j < N, wave-cooperative:
// output elements are dealt round-robin to the 64 lanes.
template <class FA, class FB, class FInit, class FOut>
__device__ __forceinline__ void wave_matmul(int M, int N, int K, FA a, FB b, FInit init, FOut out)
{
    const int total = M * N;
    for (int idx = lane_id(); idx < total; idx += kWave) {
        const int i = idx / N;
        const int j = idx - i * N;
        float s = init(i, j);
        for (int kk = 0; kk < K; ++kk) s = fmaf(a(i, kk), b(kk, j), s);
        out(i, j, s);
    }
}

// Matrix-core products for large shapes: v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulation; the same
// 64 flop/cycle/SIMD peak as the vector FMAs, but ONE instruction per 1024 multiply-adds).  The kernels that use it run
// one to three waves per SIMD out of LDS and are latency-bound on their instruction count, which this divides by ~8
// against the register-blocked loop.  NOT bit-identical to wave_matmul: the MFMA adds four products per accumulate;
// a kernel must use one variant for all its launches of a shape.
// out(i, j, init(i, j) + sum_k a(i, k) b(k, j)), i < M, j < N, on v_mfma_f32_16x16x4_f32: a "pass" is one row tile
// x CT column tiles (one a-operand read feeds CT MFMAs); passes are dealt round-robin to the block's waves.  The
// steps with all four k in range carry no clamps or masks (plain strided addresses), a last partial step is masked.
// Rows / columns beyond the matrix are clamped reads whose products land in discarded outputs.
template <int NW, int CT, class FA, class FB, class FInit, class FOut>
__device__ __forceinline__ void mfma_matmul_ct(int M, int N, int K, FA a, FB b, FInit init, FOut out)
{
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    const int lane = (int)(threadIdx.x & (kWave - 1)), li = lane & 15, lq = lane >> 4;
    const int col_groups = (N + 16 * CT - 1) / (16 * CT), passes = ((M + 15) >> 4) * col_groups;
    const int Kmain = K & ~3;
    for (int p = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); p < passes; p += NW) {
        const int ti = p / col_groups, tj = p - ti * col_groups;
        const int i0 = 16 * ti, j0 = 16 * CT * tj;
        const int ia = (i0 + li < M) ? i0 + li : M - 1;
        int ir[4], jb[CT];
        f32x4 acc[CT];
#pragma unroll
        for (int r = 0; r < 4; ++r) ir[r] = (i0 + 4 * lq + r < M) ? i0 + 4 * lq + r : M - 1;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            jb[c] = (j0 + 16 * c + li < N) ? j0 + 16 * c + li : N - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][r] = init(ir[r], jb[c]);
        }
        // chunks of four k-steps: 4 (1 + CT) operand reads issued back to back (their LDS round trips overlap),
        // then 4 CT MFMAs; the remaining full steps one at a time
        int k0 = 0;
        for (; k0 + 16 <= Kmain; k0 += 16) {
            float av[4], bv[4][CT];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                av[u] = a(ia, k0 + 4 * u + lq);
#pragma unroll
                for (int c = 0; c < CT; ++c) bv[u][c] = b(k0 + 4 * u + lq, jb[c]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][c], acc[c], 0, 0, 0);
        }
        for (; k0 < Kmain; k0 += 4) {
            const float av = a(ia, k0 + lq);
            float bv[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) bv[c] = b(k0 + lq, jb[c]);
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[c], acc[c], 0, 0, 0);
        }
        if (Kmain < K) {
            const bool kin = Kmain + lq < K;
            const int kc = kin ? Kmain + lq : K - 1;
            float av = a(ia, kc);
            av = kin ? av : 0.0f;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                float bv = b(kc, jb[c]);
                bv = kin ? bv : 0.0f;
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i0 + 4 * lq + r < M && j0 + 16 * c + li < N) out(i0 + 4 * lq + r, j0 + 16 * c + li, acc[c][r]);
    }
}

// Column tiles per pass: the smallest count whose passes fit the NW waves in one round; when none does, as fat as the
// matrix allows (these products are bound by the LDS round trip per k-step, not by the MFMA count, so fewer rounds of
// fatter passes win).  NW = 1 is the wave-per-instance form.
template <int NW = 1, class FA, class FB, class FInit, class FOut>
__device__ __forceinline__ void mfma_matmul(int M, int N, int K, FA a, FB b, FInit init, FOut out)
{
    const int rt = (M + 15) >> 4, ct = (N + 15) >> 4;
    if (ct == 1 || rt * ct <= NW) mfma_matmul_ct<NW, 1>(M, N, K, a, b, init, out);
    else if (ct == 2 || rt * ((ct + 1) >> 1) <= NW) mfma_matmul_ct<NW, 2>(M, N, K, a, b, init, out);
    else mfma_matmul_ct<NW, 4>(M, N, K, a, b, init, out);
}

template <class FA, class FB, class FInit, class FOut>
__device__ __forceinline__ void wave_matmul_mfma(int M, int N, int K, FA a, FB b, FInit init, FOut out)
{
    mfma_matmul<1>(M, N, K, a, b, init, out);
}

// f(i, j, idx) for idx = i * cols + j = lane, lane + 64, ... < rows * cols: a row-major [rows][cols] index space
// dealt round-robin to the lanes, with the (i, j) split carried incrementally -- one integer division per call
// instead of one per element (a v_rcp-based division is ~25 instructions on gfx950).
template <class F>
__device__ __forceinline__ void wave_for_2d(int rows, int cols, F f)
{
    const int total = rows * cols;
    int idx = lane_id();
    int i = idx / cols, j = idx - i * cols;
    if (idx < total) f(i, j, idx);
    if (total <= kWave) return;                 // the common case for small matrices: no stride bookkeeping at all
    const int di = kWave / cols, dj = kWave - di * cols;
    for (idx += kWave; idx < total; idx += kWave) {
        i += di;
        j += dj;
        if (j >= cols) { j -= cols; ++i; }
        f(i, j, idx);
    }
}

// Global -> LDS copy of a row-major [rows][cols] matrix into leading dimension ld.
__device__ __forceinline__ void load_matrix(float *dst, int ld, const float *src, int rows, int cols)
{
    wave_for_2d(rows, cols, [&](int r, int c, int idx) { dst[r * ld + c] = src[idx]; });
}

__device__ __forceinline__ void store_matrix(float *dst, const float *src, int ld, int rows, int cols)
{
    wave_for_2d(rows, cols, [&](int r, int c, int idx) { dst[idx] = src[r * ld + c]; });
}

// In-place Gauss-Jordan elimination of the augmented system aug[rows][width]
// (leading dimension ld) whose first `rows` columns hold the square matrix; on
// return columns rows..width-1 hold A^{-1} * RHS.
//   PIVOT = true : partial (row) pivoting -- the general inverse of lqr.py:84.
//   PIVOT = false: no pivoting; for a symmetric matrix every pivot is positive
//                  iff the matrix is positive definite, which is the failure test
//                  of tf.linalg.cholesky in ilqr.py:358.  `active` (may be null)
//                  marks rows that take part; inactive rows/cols must already
//                  be identity rows (box-QP free/clamped split).
// fac[rows] and prow[width] are LDS scratch.  Returns 0, or 1 if a pivot was
// zero (PIVOT) / non-positive or NaN (!PIVOT).  All lanes return the same value.
template <bool PIVOT>
__device__ __forceinline__ int wave_gauss_jordan(float *aug, int ld, int rows, int width, float *fac, float *prow)
{
    // Column per lane: a lane carries its column(s) j = lane, lane + 64, ... through the whole row sweep of a
    // pivot, so the only cross-lane data are the multipliers fac[i] = aug[i][p] (staged once per pivot, read
    // back as LDS broadcasts) -- no index divisions and two fences per pivot.  Element arithmetic and pivot
    // choice are those of the textbook loop: aug[i][j] <- fma(-aug[i][p], aug[piv][j] / aug[piv][p], aug[i][j]).
    (void)prow;
    const int lane = lane_id();
    int bad = 0;
    for (int p = 0; p < rows; ++p) {
        for (int i = lane; i < rows; i += kWave) fac[i] = aug[i * ld + p];
        wsync();
        int piv = p;
        if (PIVOT) {
            if (rows <= kWave) {
                // first row of maximal |entry| among rows p..rows-1, as the sequential scan finds it
                const bool in = lane >= p && lane < rows;
                const float mine = in ? fabsf(fac[lane]) : 0.0f;
                const float best = wave_max(mine);
                const unsigned long long hit = __ballot(in && mine == best);
                if (hit) piv = __ffsll((long long)hit) - 1;
            } else {
                float best = fabsf(fac[p]);
                for (int i = p + 1; i < rows; ++i) {
                    const float a = fabsf(fac[i]);
                    if (a > best) { best = a; piv = i; }
                }
            }
        }
        const float pv = fac[piv];
        if (PIVOT ? (pv == 0.0f) : !(pv > 0.0f)) bad = 1;
        const float inv = 1.0f / pv;
        for (int j = lane; j < width; j += kWave) {
            const float pr = aug[piv * ld + j] * inv;            // scaled pivot row, this column
            if (PIVOT && piv != p) aug[piv * ld + j] = aug[p * ld + j];   // row p moves into slot piv ...
            // (a chunked read-all-then-write-all row loop is faster for 16+ rows but costs the small systems of the
            // iLQR kernels 25 %; the large shapes have the block kernel)
            for (int i = 0; i < rows; ++i) {
                const float fi = fac[(PIVOT && i == piv) ? p : i];        // ... and keeps its multiplier
                const float old = aug[i * ld + j];
                aug[i * ld + j] = (i == p) ? pr : fmaf(-fi, pr, old);
            }
        }
        wsync();
    }
    return bad;
}

}  // namespace tfmpc
