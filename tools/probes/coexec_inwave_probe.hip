// Probe (gfx950): do vector instructions issued right after a matrix instruction IN THE SAME WAVE
// execute under it?  Per iteration: 4 bf16 (or f32) MFMAs, each followed by NV independent v_fma_f32.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/coexec_inwave_probe tools/probes/coexec_inwave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int F32, int NV, int WITH_M>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
    const float af = threadIdx.x * 0.001f, bf = 0.5f;
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j;
    const float m = 1.0001f, d = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (WITH_M) {
                f32x4 &c = g == 0 ? c0 : g == 1 ? c1 : g == 2 ? c2 : c3;
                if (F32) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c, 0, 0, 0);
                else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) x[v % 8] = fmaf(x[v % 8], m, d);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float acc = c0[0] + c1[1] + c2[2] + c3[3];
    for (int j = 0; j < 8; ++j) acc += x[j];
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <int F32, int NV>
void run(float *out)
{
    const int iters = 20000, blocks = 256 * 2;       // 8 waves per CU, 2 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    k<F32, NV, 1><<<blocks, 256>>>(out, 100);
    hipEventRecord(e0); k<F32, NV, 1><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[0], e0, e1);
    k<F32, NV, 0><<<blocks, 256>>>(out, 100);
    hipEventRecord(e0); k<F32, NV, 0><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[1], e0, e1);
    printf("%s mfma + %2d v_fma each: with mfma %.3f ms, vector ops alone %.3f ms\n", F32 ? "f32 " : "bf16", NV, ms[0], ms[1]);
}

int main()
{
    float *out;
    hipMalloc(&out, 4096);
    run<0, 0>(out); run<0, 2>(out); run<0, 4>(out); run<0, 6>(out); run<0, 8>(out); run<0, 12>(out); run<0, 16>(out);
    run<1, 0>(out); run<1, 4>(out); run<1, 8>(out); run<1, 16>(out);
    return 0;
}
