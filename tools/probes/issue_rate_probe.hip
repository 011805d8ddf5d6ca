// Probe (gfx950): issue cost in cycles of single wave64 instructions, by timing long runs of one
// instruction kind (inline asm, 4 independent registers rotating) with W waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/issue_rate_probe tools/probes/issue_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP4(a, b, c, d) a b c d
#define REP16(x0, x1, x2, x3) REP4(x0, x1, x2, x3) REP4(x0, x1, x2, x3) REP4(x0, x1, x2, x3) REP4(x0, x1, x2, x3)

#define KERNEL(NAME, I0, I1, I2, I3)                                                              \
    __global__ __launch_bounds__(1024) void NAME(float *out, int iters)                            \
    {                                                                                              \
        float v0 = threadIdx.x, v1 = 1.5f, v2 = 2.5f, v3 = 3.5f, v4 = 1.0001f, v5 = 0.5f;          \
        float w0 = 4.f, w1 = 5.f, w2 = 6.f, w3 = 7.f;                                              \
        for (int it = 0; it < iters; ++it) {                                                       \
            asm volatile(REP16(I0 "\n", I1 "\n", I2 "\n", I3 "\n")                                    \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) \
                         : "v"(v4), "v"(v5)                                                         \
                         : "s20", "s21", "s22", "s23");                                            \
        }                                                                                          \
        if (v0 + v1 + v2 + v3 + w0 + w1 + w2 + w3 == 123.456f) out[threadIdx.x] = v0;              \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9", "v_fma_f32 %2, %2, %8, %9", "v_fma_f32 %3, %3, %8, %9")
KERNEL(k_mul, "v_mul_f32 %0, %0, %8", "v_mul_f32 %1, %1, %8", "v_mul_f32 %2, %2, %8", "v_mul_f32 %3, %3, %8")
KERNEL(k_add, "v_add_f32 %0, %0, %8", "v_add_f32 %1, %1, %8", "v_add_f32 %2, %2, %8", "v_add_f32 %3, %3, %8")
KERNEL(k_mov, "v_mov_b32 %0, %4", "v_mov_b32 %1, %5", "v_mov_b32 %2, %6", "v_mov_b32 %3, %7")
KERNEL(k_and, "v_and_b32 %0, 0xffff0000, %0", "v_and_b32 %1, 0xffff0000, %1", "v_and_b32 %2, 0xffff0000, %2", "v_and_b32 %3, 0xffff0000, %3")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0", "v_lshlrev_b32 %1, 1, %1", "v_lshlrev_b32 %2, 1, %2", "v_lshlrev_b32 %3, 1, %3")
KERNEL(k_xor, "v_xor_b32 %0, %0, %4", "v_xor_b32 %1, %1, %5", "v_xor_b32 %2, %2, %6", "v_xor_b32 %3, %3, %7")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %4, vcc", "v_cndmask_b32 %1, %1, %5, vcc", "v_cndmask_b32 %2, %2, %6, vcc", "v_cndmask_b32 %3, %3, %7, vcc")
KERNEL(k_cvtpk, "v_cvt_pk_bf16_f32 %0, %4, %5", "v_cvt_pk_bf16_f32 %1, %5, %6", "v_cvt_pk_bf16_f32 %2, %6, %7", "v_cvt_pk_bf16_f32 %3, %7, %4")
KERNEL(k_rcp, "v_rcp_f32 %0, %0", "v_rcp_f32 %1, %1", "v_rcp_f32 %2, %2", "v_rcp_f32 %3, %3")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3", "v_readlane_b32 s21, %1, 5", "v_readlane_b32 s22, %2, 7", "v_readlane_b32 s23, %3, 9")
KERNEL(k_readlane_fma, "v_readlane_b32 s20, %0, 3", "v_fma_f32 %1, s21, %1, %9", "v_readlane_b32 s21, %2, 7", "v_fma_f32 %3, s20, %3, %9")
KERNEL(k_dppmov, "v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %5 row_half_mirror row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %2, %6 row_ror:4 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %7 row_bcast:15 row_mask:0xf bank_mask:0xf")
KERNEL(k_dppadd, "v_add_f32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %1, %5, %1 row_half_mirror row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %2, %6, %2 row_ror:4 row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %3, %7, %3 row_mirror row_mask:0xf bank_mask:0xf")
KERNEL(k_permswap, "v_permlane16_swap_b32 %0, %4", "v_permlane32_swap_b32 %1, %5", "v_permlane16_swap_b32 %2, %6", "v_permlane32_swap_b32 %3, %7")
KERNEL(k_swizzle, "ds_swizzle_b32 %0, %4 offset:0x801f", "ds_swizzle_b32 %1, %5 offset:0x801f", "ds_swizzle_b32 %2, %6 offset:0x801f", "ds_swizzle_b32 %3, %7 offset:0x801f\ns_waitcnt lgkmcnt(0)")
KERNEL(k_bperm, "ds_bpermute_b32 %0, %4, %5", "ds_bpermute_b32 %1, %5, %6", "ds_bpermute_b32 %2, %6, %7", "ds_bpermute_b32 %3, %7, %4\ns_waitcnt lgkmcnt(0)")

typedef __attribute__((ext_vector_type(2))) float f32x2;
#define KERNEL2(NAME, I0, I1, I2, I3)                                                             \
    __global__ __launch_bounds__(1024) void NAME(float *out, int iters)                            \
    {                                                                                              \
        f32x2 v0 = {(float)threadIdx.x, 1.f}, v1 = {1.5f, 2.f}, v2 = {2.5f, 3.f}, v3 = {3.5f, 4.f}; \
        f32x2 v4 = {1.0001f, 1.0002f}, v5 = {0.5f, 0.25f};                                          \
        for (int it = 0; it < iters; ++it) {                                                       \
            asm volatile(REP16(I0 "\n", I1 "\n", I2 "\n", I3 "\n")                                    \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)                                  \
                         : "v"(v4), "v"(v5));                                                       \
        }                                                                                          \
        if (v0[0] + v1[1] + v2[0] + v3[1] == 123.456f) out[threadIdx.x] = v0[0];                   \
    }
KERNEL2(k_pkfma, "v_pk_fma_f32 %0, %0, %4, %5", "v_pk_fma_f32 %1, %1, %4, %5", "v_pk_fma_f32 %2, %2, %4, %5", "v_pk_fma_f32 %3, %3, %4, %5")
KERNEL2(k_pkadd, "v_pk_add_f32 %0, %0, %4", "v_pk_add_f32 %1, %1, %4", "v_pk_add_f32 %2, %2, %4", "v_pk_add_f32 %3, %3, %4")
KERNEL2(k_pkmul, "v_pk_mul_f32 %0, %0, %4", "v_pk_mul_f32 %1, %1, %4", "v_pk_mul_f32 %2, %2, %4", "v_pk_mul_f32 %3, %3, %4")
KERNEL2(k_mov64, "v_mov_b64 %0, %4", "v_mov_b64 %1, %5", "v_mov_b64 %2, %4", "v_mov_b64 %3, %5")

template <class K>
void run(const char *name, K kern, float *out, int ninstr_per_iter)
{
    const int iters = 60000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-22s", name);
    for (int waves_per_simd : {1, 2, 4, 4}) {
        const int threads = 64 * 4 * waves_per_simd;       // one block per CU
        kern<<<256, threads>>>(out, iters);
        hipEventRecord(e0);
        kern<<<256, threads>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * ninstr_per_iter * waves_per_simd;
        printf("  %dw/SIMD: %6.2f ns/instr", waves_per_simd, ms * 1e6 / instr_per_simd);
    }
    printf("\n");
}

int main()
{
    float *out;
    hipMalloc(&out, 1 << 16);
    run("v_fma_f32", k_fma, out, 16);
    run("v_mul_f32", k_mul, out, 16);
    run("v_add_f32", k_add, out, 16);
    run("v_pk_fma_f32", k_pkfma, out, 16);
    run("v_pk_add_f32", k_pkadd, out, 16);
    run("v_pk_mul_f32", k_pkmul, out, 16);
    run("v_mov_b64", k_mov64, out, 16);
    run("v_mov_b32", k_mov, out, 16);
    run("v_and_b32", k_and, out, 16);
    run("v_lshlrev_b32", k_lshl, out, 16);
    run("v_xor_b32", k_xor, out, 16);
    run("v_cndmask_b32", k_cndmask, out, 16);
    run("v_cvt_pk_bf16_f32", k_cvtpk, out, 16);
    run("v_rcp_f32", k_rcp, out, 16);
    run("v_readlane_b32", k_readlane, out, 16);
    run("readlane,fma pairs", k_readlane_fma, out, 16);
    run("v_mov_b32_dpp", k_dppmov, out, 16);
    run("v_add_f32_dpp", k_dppadd, out, 16);
    run("v_permlane{16,32}_swap", k_permswap, out, 16);
    run("ds_swizzle_b32", k_swizzle, out, 16);
    run("ds_bpermute_b32", k_bperm, out, 16);
    return 0;
}
