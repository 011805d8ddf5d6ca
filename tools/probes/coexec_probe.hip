// Probe (gfx950): how much do matrix-core instructions of one wave overlap with vector instructions
// of ANOTHER wave on the same SIMD?  A workgroup of 8 waves puts 2 waves on each SIMD; waves 0-3
// run a matrix-op loop, waves 4-7 a vector-op loop; each role is also timed alone.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/coexec_probe tools/probes/coexec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MKIND, int VKIND>
__global__ __launch_bounds__(512) void k(float *out, int iters, int run_m, int run_v)
{
    const int wave = threadIdx.x >> 6;
    float acc = 0.f;
    if (wave < 4) {
        if (!run_m) return;
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        if (MKIND == 0) {            // bf16 16x16x32, 4 independent chains
            bf16x8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
            for (int it = 0; it < iters; ++it) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
            }
        } else if (MKIND == 1) {     // bf16, one dependent chain
            bf16x8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
            for (int it = 0; it < iters; ++it) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            }
        } else {                     // f32 16x16x4, 4 independent chains
            const float a = threadIdx.x * 0.001f, b = 0.5f;
            for (int it = 0; it < iters; ++it) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            }
        }
        acc = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        if (!run_v) return;
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
        const float m = 1.0001f, d = 0.5f;
        if (VKIND == 0) {            // 16 independent-ish v_fma_f32 per iteration
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x0 = fmaf(x0, m, d); x1 = fmaf(x1, m, d); x2 = fmaf(x2, m, d); x3 = fmaf(x3, m, d);
                }
            }
        } else if (VKIND == 1) {     // 16 v_pk_fma_f32
            f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x1, x0}, p3 = {x3, x2};
            const f32x2 mm = {m, m}, dd = {d, d};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    p0 = __builtin_elementwise_fma(p0, mm, dd); p1 = __builtin_elementwise_fma(p1, mm, dd);
                    p2 = __builtin_elementwise_fma(p2, mm, dd); p3 = __builtin_elementwise_fma(p3, mm, dd);
                }
            }
            x0 = p0[0] + p1[1] + p2[0] + p3[1];
        } else if (VKIND == 3) {     // 16 integer ops (v_xor / v_add_u32 / v_lshl_add)
            unsigned u0 = threadIdx.x, u1 = 7, u2 = 9, u3 = 11;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u0 = (u0 ^ u1) + 0x9e3779b9u; u1 = (u1 << 3) + u2; u2 = (u2 ^ u3) + 0x7f4a7c15u; u3 = (u3 << 5) + u0;
                }
            }
            x0 = u0 + u1 + u2 + u3;
        } else if (VKIND == 4) {     // 8 x (v_cvt_pk_bf16_f32 + shift back)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                    const bf16x2 h = {(__bf16)x0, (__bf16)x1};
                    const unsigned hb = __builtin_bit_cast(unsigned, h);
                    x0 = __uint_as_float((hb << 16) ^ 0x00010000u);
                    x1 = __uint_as_float((hb & 0xffff0000u) ^ 0x00010000u);
                }
            }
        } else if (VKIND == 5) {     // 16 v_readlane only (results summed on the scalar unit)
            int sacc = 0;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 16; ++j) sacc ^= __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1) + it, j + 3) + j;
            }
            x0 = sacc;
        } else {                     // 8 x (v_readlane + v_fma with the scalar)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1), j + 3));
                    x0 = fmaf(s, x0, d);
                    x1 = x1 * m;
                }
            }
        }
        acc = x0 + x1 + x2 + x3;
    }
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <int MK, int VK>
void run(const char *name, float *out)
{
    const int iters = 20000, blocks = 256 * 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[3];
    const int cfg[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int c = 0; c < 3; ++c) {
        k<MK, VK><<<blocks, 512>>>(out, 100, cfg[c][0], cfg[c][1]);
        hipEventRecord(e0);
        k<MK, VK><<<blocks, 512>>>(out, iters, cfg[c][0], cfg[c][1]);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[c], e0, e1);
    }
    printf("%-44s matrix alone %.3f ms, vector alone %.3f ms, both %.3f ms  -> overlap %.0f %% of the shorter\n", name, ms[0], ms[1], ms[2],
           100.0 * (ms[0] + ms[1] - ms[2]) / (ms[0] < ms[1] ? ms[0] : ms[1]));
}

int main()
{
    float *out;
    hipMalloc(&out, 4096);
    run<0, 0>("bf16 mfma x4 indep   | v_fma_f32", out);
    run<0, 1>("bf16 mfma x4 indep   | v_pk_fma_f32", out);
    run<0, 2>("bf16 mfma x4 indep   | v_readlane+v_fma", out);
    run<1, 0>("bf16 mfma dependent  | v_fma_f32", out);
    run<1, 2>("bf16 mfma dependent  | v_readlane+v_fma", out);
    run<0, 3>("bf16 mfma x4 indep   | integer ops", out);
    run<0, 4>("bf16 mfma x4 indep   | v_cvt_pk_bf16 + shifts", out);
    run<0, 5>("bf16 mfma x4 indep   | v_readlane only", out);
    run<2, 3>("f32 mfma x4 indep    | integer ops", out);
    run<2, 0>("f32 mfma x4 indep    | v_fma_f32", out);
    run<2, 1>("f32 mfma x4 indep    | v_pk_fma_f32", out);
    return 0;
}
