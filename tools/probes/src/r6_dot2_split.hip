// Is `x - bf16(x)` ONE instruction on gfx950?  v_dot2c_f32_bf16 D += a.lo * b.lo + a.hi * b.hi with b = (-1, 0) or (0, -1) picks one half of a
// packed bf16 pair and subtracts it from an fp32 value -- the residual step of the bf16x3 split (mfma_bf16x3.h) without the shift / mask that turns
// the bf16 back into an fp32 first: 14 instead of 22 vector instructions per split of four values.  Measured (round 6):
//  (1) EXACT: bit-identical to the two-instruction form on 2^26 random bit patterns incl. denormals; the only differences are pairs whose OTHER half
//      rounded to a bf16 infinity (|x| >= 3.39e38: 0 x inf).
//  (2) the selector must reach the instruction in a register: given the constant 0x0000bf80 the compiler emits the INLINE constant -1.0, which the
//      instruction reads as the 32-bit pattern 0xbf800000 -- the other half (that form differed on every value).
//  (3) HALF RATE: a launch that fills the chip takes 1.21 - 1.23 x as long with four of them as with the six instructions they replace (two shifts /
//      masks + four subtractions), i.e. ~7.4 cycles each, plus three wait states before the result can be used.  In the kernels: headline 1.700 ->
//      1.686 ms (within a box's noise), iLQR API 3.45 -> 3.45, LQR n = 32 1.236 -> 1.279 ms.  NOT adopted.
//   hipcc -O3 --offload-arch=gfx950 -o build_tmp/r6_dot2 tools/probes/src/r6_dot2_split.hip && build_tmp/r6_dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
__device__ __forceinline__ unsigned pack_bf16(float a, float b) { const bf16x2 p = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(unsigned, p); }
__device__ __forceinline__ float dot2(unsigned p, unsigned s, float c)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, s), c, false);     // v_dot2c_f32_bf16
}
// (in a register the compiler cannot see through: given the constant 0x0000bf80 it emits the inline constant -1.0, which the instruction reads as
// the 32-bit pattern 0xbf800000 -- the other half of the pair; that form differed on every value)
__device__ __forceinline__ unsigned select_lo() { unsigned s; asm("s_mov_b32 %0, 0xbf80" : "=s"(s)); return s; }
__device__ __forceinline__ unsigned select_hi() { unsigned s; asm("s_mov_b32 %0, 0xbf800000" : "=s"(s)); return s; }
#define kLo select_lo()
#define kHi select_hi()
__global__ void check(const float *x, size_t n, unsigned long long *bad, float *first)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    const unsigned h = pack_bf16(a, b);
    const float r0 = a - __uint_as_float(h << 16), r1 = b - __uint_as_float(h & 0xffff0000u);
    const float d0 = dot2(h, kLo, a), d1 = dot2(h, kHi, b);
    const unsigned m = pack_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xffff0000u);
    const float e0 = dot2(m, kLo, r0), e1 = dot2(m, kHi, r1);
    const bool ok = __float_as_uint(r0) == __float_as_uint(d0) && __float_as_uint(r1) == __float_as_uint(d1) && __float_as_uint(q0) == __float_as_uint(e0) &&
                    __float_as_uint(q1) == __float_as_uint(e1);
    if (!ok) atomicAdd(bad + 1, (fabsf(a) < 3.38e38f && fabsf(b) < 3.38e38f) ? 1ull : 0ull);     // (a partner that rounds to a bf16 infinity: 0 x inf)
    if (!ok && atomicAdd(bad, 1ull) == 0) { first[0] = a; first[1] = b; first[2] = r0; first[3] = d0; first[4] = r1; first[5] = d1; first[6] = q0; first[7] = e0; }
}
template <int MODE>
__global__ void rate(float *out, long long *ticks, int reps)
{
    float a0 = threadIdx.x * 1.0001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    unsigned p = pack_bf16(a0, a1);
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            asm volatile("" : "+v"(p));                               // (the shifts are per value in the split: not hoisted)
            if (MODE == 0) {
                a0 = a0 - __uint_as_float(p << 16); a1 = a1 - __uint_as_float(p & 0xffff0000u);
                a2 = a2 - __uint_as_float(p << 16); a3 = a3 - __uint_as_float(p & 0xffff0000u);
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else {
                a0 = dot2(p, kLo, a0); a1 = dot2(p, kHi, a1); a2 = dot2(p, kLo, a2); a3 = dot2(p, kHi, a3);
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[MODE] = t1 - t0;
}
int main()
{
    const size_t n = (size_t)1 << 26;
    std::vector<float> h(n);
    srand(7);
    for (size_t i = 0; i < n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
        if (((u >> 23) & 0xff) == 0xff) u &= ~(1u << 30);          // finite
        memcpy(&h[i], &u, 4);
    }
    const float specials[] = {0.f, -0.f, 1.f, -1.f, 1.17549435e-38f, 1e-40f, -1e-40f, 3.4028235e38f, -3.4028235e38f, 3.3895314e38f, 1.0039062f, 1.00390625f, 0.99609375f};
    for (size_t i = 0; i < sizeof(specials) / 4; ++i) h[i] = specials[i];
    float *x, *first, *out; unsigned long long *bad; long long *ticks;
    if (hipMalloc(&x, n * 4) || hipMalloc(&first, 32) || hipMalloc(&bad, 16) || hipMalloc(&out, 16 << 20) || hipMalloc(&ticks, 16)) return 2;
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 16);
    check<<<(unsigned)(n / 2 / 256), 256>>>(x, n, bad, first);
    unsigned long long nbs[2]; float f[8];
    hipMemcpy(nbs, bad, 16, hipMemcpyDeviceToHost); const unsigned long long nb = nbs[1]; hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
    printf("values %zu  differing pairs %llu, of which with both values below 3.38e38 (no bf16 infinity in the pair): %llu\n", n, nbs[0], nb);
    if (nb) printf("first: a %.9g b %.9g | r0 %.9g dot %.9g | r1 %.9g dot %.9g | q0 %.9g dot %.9g\n", f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
    for (int waves = 1; waves <= 4; waves += 3) {
        long long t[2];
        const int reps = 2000;
        float ms[2] = {0.f, 0.f};
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it = 0; it < 2; ++it) {
            // 2 048 blocks: eight rounds of one block per CU when a block is sixteen waves; the event pair times the whole launch (throughput),
            // clock64 one wave's own time (latency)
            hipEventRecord(e0); rate<0><<<2048, 256 * waves>>>(out, ticks, reps); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
            hipEventRecord(e0); rate<1><<<2048, 256 * waves>>>(out, ticks, reps); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
            if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 3; }
        }
        hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost);
        const double per = 1.0 / (reps * 16.0 * 4.0);
        printf("waves per SIMD %d: shift/mask + v_sub_f32 %.2f ticks per residual, v_dot2c_f32_bf16 %.2f  (clock64 ticks of one wave);  whole launch %.3f / %.3f ms\n", waves,
               t[0] * per, t[1] * per, ms[0], ms[1]);
    }
    return nb != 0;
}
