// lqr_kernels.h -- internal launch interface shared by the LQR kernel variants and
// the C-ABI dispatcher (lqr_dispatch.hip).  Not installed; the public contract is
// include/tfmpc_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/tfmpc_hip.h"

namespace tfmpc {

// gfx950: 160 KiB of LDS per CU, all of it addressable by one workgroup.
constexpr size_t kMaxLdsBytes = 160 * 1024;

// fp32 -> bf16, round to nearest even on the bits (NaN stays NaN: the quiet bit survives the shift)
__host__ __device__ inline uint16_t lqr_to_bf16(float x)
{
    unsigned u = __builtin_bit_cast(unsigned, x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

struct LqrArgs {
    int B, n, m, T;
    const float *F, *f, *C, *c, *x0;
    long sF, sf, sC, sc;          // batch strides (elements), 0 = shared
    float *K, *k;                 // [B][T][m][n], [B][T][m]  (read by forward-only launches)
    long sK, sk;                  // batch strides of K, k
    float *V, *v, *cst;           // optional value-function outputs
    // optional 16-bit (bf16, round to nearest even) copies of the policy / value-function outputs, in the layouts of
    // K, k, V, v, cst (tfmpc_lqr_*_bf16out_f32; served by the matrix-core kernels and the wave kernel)
    uint16_t *K16, *k16, *V16, *v16, *cst16;
    float *states, *actions, *costs;
    int32_t *status;
};

size_t lqr_generic_smem_bytes(int n, int m);
int lqr_generic_launch(const LqrArgs &a, bool backward, bool forward, hipStream_t stream);

// Workgroup-per-instance variant for large shapes (lqr_block.hip): four waves, products on the f32 matrix cores.
size_t lqr_block_smem_bytes(int n, int m);
int lqr_block_launch(const LqrArgs &a, bool backward, bool forward, hipStream_t stream);

// Lane-per-instance variant for tiny shapes, n + m <= 6 (lqr_lane.hip).
bool lqr_lane_supported(int n, int m);
int lqr_lane_launch(const LqrArgs &a, bool backward, bool forward, hipStream_t stream);

// MFMA variant for shapes up to n = 32, m = 16 beyond the 16 x 8 tile (lqr_mfma32x16.hip): 2 x 2 tiles of bf16x3.
bool lqr_mfma32_supported(int n, int m);
int lqr_mfma32_launch(const LqrArgs &a, bool backward, bool forward, hipStream_t stream);

// MFMA variant for the BASELINE.json headline shape (lqr_mfma16x8.hip).
bool lqr_mfma_supported(int n, int m);
int lqr_mfma_launch(const LqrArgs &a, bool backward, bool forward, hipStream_t stream);

}  // namespace tfmpc
