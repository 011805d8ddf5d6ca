// block_ops.h -- building blocks for "one WORKGROUP of NW wavefronts owns one problem instance": the
// large-shape counterpart of wave_ops.h.  Above n + m ~ 24 a single wave per instance is latency-bound on its
// own instruction stream (one wave per SIMD fits in LDS, every LDS round trip is exposed); NW = 4 waves share
// the instance's LDS tiles, split every phase NW ways and leave 3 workgroups = 3 waves per SIMD to overlap.
// gfx950 only.  __syncthreads() is a real s_barrier here.
#pragma once

#include <hip/hip_runtime.h>

#include "wave_ops.h"

namespace tfmpc {

template <int NW>
struct Block {
    static constexpr int kThreads = NW * kWave;
    __device__ static __forceinline__ int tid() { return threadIdx.x; }
    __device__ static __forceinline__ int lane() { return threadIdx.x & (kWave - 1); }
    // wave index as a scalar: uniform loops over it then run on the SALU
    __device__ static __forceinline__ int wave() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
};

// f(i, j, idx) over a row-major [rows][cols] index space dealt round-robin to the block's threads
// (incremental (i, j) split: one division per call, see wave_for_2d).
template <int NW, class F>
__device__ __forceinline__ void block_for_2d(int rows, int cols, F f)
{
    constexpr int NT = NW * kWave;
    const int total = rows * cols;
    int idx = threadIdx.x;
    int i = idx / cols, j = idx - i * cols;
    const int di = NT / cols, dj = NT - di * cols;
    for (; idx < total; idx += NT) {
        f(i, j, idx);
        i += di;
        j += dj;
        if (j >= cols) { j -= cols; ++i; }
    }
}

// Sum over the block; scratch = NW floats of LDS.  Two barriers (the second frees the scratch for the next call).
template <int NW>
__device__ __forceinline__ float block_sum(float v, float *scratch)
{
    v = wave_sum(v);
    if (NW == 1) return v;
    if (Block<NW>::lane() == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += scratch[w];
    __syncthreads();
    return s;
}

// block_matmul_mfma<NW>: the pass-based matrix-core product of wave_ops.h (mfma_matmul) with the passes dealt to
// the block's NW waves.
template <int NW, class FA, class FB, class FInit, class FOut>
__device__ __forceinline__ void block_matmul_mfma(int M, int N, int K, FA a, FB b, FInit init, FOut out)
{
    mfma_matmul<NW>(M, N, K, a, b, init, out);
}

// Gauss-Jordan WITHOUT pivoting of an m x width system, m <= 16, width <= 64, by ONE wave entirely in registers:
// lane j holds column j (reg[i] = entry (i, j)), the multipliers reach all lanes through v_readlane, no LDS and no
// barriers.  ~45 instructions per pivot against three barriers and six LDS round trips of block_gauss_jordan.
// Arithmetic per element is wave_gauss_jordan<false>'s: row_s *= 1 / pivot; row_p = fma(-a_ps, row_s, row_p).
// On return reg[i] of lane j >= m is (A^-1 RHS)(i, j - m).  Returns 0, 1 (zero pivot) or 2 (negative / NaN pivot:
// the matrix is not positive definite).
template <int MM, class Load>
__device__ __forceinline__ int wave_gj_registers(int m, int width, Load load, float (&reg)[16])
{
    const int lane = lane_id();
    const int j = lane < width ? lane : width - 1;
#pragma unroll
    for (int i = 0; i < MM; ++i) reg[i] = load(i < m ? i : m - 1, j);
#pragma unroll
    for (int i = 0; i < MM; ++i) reg[i] = (i < m) ? reg[i] : 0.0f;
    int bad = 0;
#pragma unroll
    for (int s = 0; s < MM; ++s) {
        // steps s >= m (at most three: MM is m rounded up to a multiple of four) run as arithmetic no-ops
        const bool on = s < m;
        const float pv = on ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, reg[s]), s)) : 1.0f;
        if (!(pv > 0.0f)) bad |= (pv == 0.0f) ? 1 : 2;
        float inv = __builtin_amdgcn_rcpf(pv);                 // v_rcp_f32 (1 ulp) + one Newton step: the pivot's
        inv = fmaf(fmaf(-pv, inv, 1.0f), inv, inv);            // reciprocal sits on the critical path of every step
        reg[s] *= inv;
#pragma unroll
        for (int p = 0; p < MM; ++p) {
            if (p == s) continue;
            float l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, reg[p]), s));
            l = on ? l : 0.0f;
            reg[p] = fmaf(-l, reg[s], reg[p]);
        }
    }
    return bad;
}

template <class Load>
__device__ __forceinline__ int wave_gj16_registers(int m, int width, Load load, float (&reg)[16])
{
    if (m <= 4) return wave_gj_registers<4>(m, width, load, reg);
    if (m <= 8) return wave_gj_registers<8>(m, width, load, reg);
    if (m <= 12) return wave_gj_registers<12>(m, width, load, reg);
    return wave_gj_registers<16>(m, width, load, reg);
}

// y(r) = init(r) + sum_k a(r, k) x(k), r < R: rows dealt as r = tid % RP (RP = power of two >= R, at most the
// block), the k range cut into NT / RP interleaved slices whose partial sums meet in `scratch` (NT floats of
// LDS).  Two barriers per pass.
template <int NW, class FA, class FX, class FInit, class FOut>
__device__ __forceinline__ void block_matvec(int R, int K, FA a, FX x, FInit init, float *scratch, FOut out)
{
    constexpr int NT = NW * kWave;
    int rp_log = 0;
    while ((1 << rp_log) < R && (1 << rp_log) < NT) ++rp_log;
    const int RP = 1 << rp_log, S = NT >> rp_log;
    const int tid = threadIdx.x, rl = tid & (RP - 1), sl = tid >> rp_log;
    for (int r0 = 0; r0 < R; r0 += RP) {
        const int r = r0 + rl;
        float acc = 0.0f;
        if (r < R)
            for (int k = sl; k < K; k += S) acc = fmaf(a(r, k), x(k), acc);
        scratch[tid] = acc;                               // [slice][RP]
        __syncthreads();
        if (tid < RP && r < R) {
            float y = init(r);
            for (int s = 0; s < S; ++s) y += scratch[(s << rp_log) + tid];
            out(r, y);
        }
        __syncthreads();
    }
}

// Gauss-Jordan of aug[rows][width] (wave_gauss_jordan's contract and arithmetic) by a block: wave w sweeps rows
// w, w + NW, ...; columns go to lanes.  Per pivot: the multiplier column is staged in fac[], the scaled pivot row
// and the row it displaces in prow[] / rowp[] (width floats each), so that every element of aug is read and
// written by exactly one thread between barriers.  Three barriers per pivot.
template <bool PIVOT, int NW>
__device__ __forceinline__ int block_gauss_jordan(float *aug, int ld, int rows, int width, float *fac, float *prow,
                                                  float *rowp)
{
    constexpr int NT = NW * kWave;
    const int tid = threadIdx.x, lane = Block<NW>::lane(), wave = Block<NW>::wave();
    int bad = 0;
    for (int p = 0; p < rows; ++p) {
        for (int i = tid; i < rows; i += NT) fac[i] = aug[i * ld + p];
        __syncthreads();
        int piv = p;
        if (PIVOT) {
            if (rows <= kWave) {            // every wave finds the same first row of maximal |entry|
                const bool in = lane >= p && lane < rows;
                const float mine = in ? fabsf(fac[lane]) : 0.0f;
                const float best = wave_max(mine);
                const unsigned long long hit = __ballot(in && mine == best);
                if (hit) piv = __ffsll((long long)hit) - 1;
            } else {
                float best = fabsf(fac[p]);
                for (int i = p + 1; i < rows; ++i) {
                    const float v = fabsf(fac[i]);
                    if (v > best) { best = v; piv = i; }
                }
            }
        }
        const float pv = fac[piv];
        if (PIVOT ? (pv == 0.0f) : !(pv > 0.0f)) bad = 1;
        const float inv = 1.0f / pv;
        for (int j = tid; j < width; j += NT) {
            prow[j] = aug[piv * ld + j] * inv;
            rowp[j] = aug[p * ld + j];
        }
        __syncthreads();
        for (int j = lane; j < width; j += kWave) {
            const float pr = prow[j], rp = rowp[j];
            for (int i0 = wave; i0 < rows; i0 += 4 * NW) {       // four of this wave's rows at a time, reads first
                float fi[4], old[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = (i0 + r * NW < rows) ? i0 + r * NW : rows - 1;
                    fi[r] = fac[(PIVOT && i == piv) ? p : i];       // the displaced row keeps its multiplier
                    old[r] = (PIVOT && i == piv) ? rp : aug[i * ld + j];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + r * NW;
                    if (i < rows) aug[i * ld + j] = (i == p) ? pr : fmaf(-fi[r], pr, old[r]);
                }
            }
        }
        __syncthreads();
    }
    return bad;
}

}  // namespace tfmpc
