// matvec_bf16x3_probe.hip -- what does the fp32-accurate 32 x 32 coupling product of the cfg5 kernels cost in isolation?
// One step of ilqr_adjoint_mfma_kernel's HVAC rollout chain is: split the 8 state rows a lane holds into three bf16 parts
// (~44 vector instructions), 12 v_mfma_f32_16x16x32_bf16 (two output tiles x six part pairs, operands resident), a few
// packed FMAs on the result, which is the next step's input (a dependent chain).  In the solver this block accounts for
// ~690 SIMD cycles per chain-step (tools/probes/cfg5_phases.py with -DTFMPC_PROBE_NO_MATRIX_PRODUCT), against
// 12 x 16 + 44 x ~4 = 370 on paper.  Variants: 0 = split + MFMA chain (as the kernel), 1 = MFMA chain only (parts reused),
// 2 = split only, 3 = as 0 with TWO independent chains per wave (the kernel's two step sizes per pass).
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o tools/probes/matvec_probe tools/probes/matvec_bf16x3_probe.hip
// Run:   tools/probes/matvec_probe            (prints cycles per chain-step per wave at 1 and 2 waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ unsigned pack(float a, float b)
{
    const bf16x2 p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}
struct Parts { u32x4 h, m, l; };
__device__ __forceinline__ Parts split(const float (&z)[8])
{
    Parts P;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = z[2 * k], b = z[2 * k + 1];
        const unsigned h = pack(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const unsigned m = pack(ra, rb);
        const unsigned l = pack(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
        P.h[k] = h; P.m[k] = m; P.l[k] = l;
    }
    return P;
}
__device__ __forceinline__ f32x4 mf(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int VARIANT>
__global__ __launch_bounds__(64) void probe(const float *in, float *out, int steps, long long *cycles)
{
    constexpr int NC = VARIANT == 3 ? 2 : 1;
    const int lane = threadIdx.x;
    u32x4 Ah[2], Am[2], Al[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = in[(lane * 16 + a * 8 + e) & 1023] * 0.01f;
        const Parts P = split(v);
        Ah[a] = P.h; Am[a] = P.m; Al[a] = P.l;
    }
    float x[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) x[c][e] = in[(lane + 64 * e + 7 * c) & 1023];
    Parts Z0 = split(x[0]);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            Parts Z = Z0;
            if (VARIANT != 1) Z = split(x[c]);
            f32x4 acc[2] = {{x[c][0], x[c][1], x[c][2], x[c][3]}, {x[c][4], x[c][5], x[c][6], x[c][7]}};
            if (VARIANT != 2) {
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = mf(Ah[a], Z.h, acc[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = mf(Ah[a], Z.m, acc[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = mf(Am[a], Z.h, acc[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = mf(Am[a], Z.m, acc[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = mf(Ah[a], Z.l, acc[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = mf(Al[a], Z.h, acc[a]);
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][r] += __uint_as_float(Z.h[r]) + __uint_as_float(Z.m[r]) + __uint_as_float(Z.l[r]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[c][4 * a + r] = fmaf(acc[a][r], 0.25f, x[c][4 * a + r] * 0.5f);      // the next step's input
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += x[c][e];
    out[blockIdx.x * 64 + lane] = sum;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char *name, int chains)
{
    const int steps = 2000;
    float *in, *out;
    long long *cyc;
    hipMalloc(&in, 1024 * 4); hipMalloc(&out, 4096 * 64 * 4); hipMalloc(&cyc, 4096 * 8);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int blocks : {1, 1024, 2048, 4096}) {
        hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(64), 0, 0, in, out, steps, cyc);
        hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(64), 0, 0, in, out, steps, cyc);
        hipDeviceSynchronize();
        static long long hc[4096];
        hipMemcpy(hc, cyc, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < blocks; ++i) s += (double)hc[i];
        printf("%-34s %4.1f waves per SIMD: %7.0f cycles per chain-step per wave\n", name, blocks / 1024.0, s / blocks / steps / chains);
    }
    hipFree(in); hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("split + 12 MFMA (one chain)", 1);
    run<1>("12 MFMA only", 1);
    run<2>("split only", 1);
    run<3>("split + 12 MFMA, two chains", 2);
    return 0;
}
