// Probe (gfx950): is  r = x - bf16(x)  computed exactly by v_dot2c_f32_bf16 (r = h * -1 + 0 * h' + x)?
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/dot2_probe tools/probes/dot2_split_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void probe(const float *x, float *r_dot, float *r_sub, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    const bf16x2 h = {(__bf16)x0, (__bf16)x1};
    const bf16x2 s0 = {(__bf16)-1.0f, (__bf16)0.0f};
    const bf16x2 s1 = {(__bf16)0.0f, (__bf16)-1.0f};
    r_dot[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(h, s0, x0, false);
    r_dot[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(h, s1, x1, false);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    r_sub[2 * i] = x0 - __uint_as_float(hb << 16);
    r_sub[2 * i + 1] = x1 - __uint_as_float(hb & 0xffff0000u);
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> x(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned b = ((unsigned)rand() << 16) ^ (unsigned)rand();
        if (i % 4 == 0) { float f = (float)rand() / RAND_MAX * 20.f - 10.f; memcpy(&b, &f, 4); }       // ordinary values
        if (i % 4 == 1) { float f = ((float)rand() / RAND_MAX - 0.5f) * 1e-3f; memcpy(&b, &f, 4); }    // residual-sized
        unsigned e = (b >> 23) & 0xff;
        if (e == 0xff) b &= 0x7f7fffffu;                                                                // no inf/nan
        memcpy(&x[i], &b, 4);
    }
    float *dx, *dd, *ds;
    hipMalloc(&dx, n * 4); hipMalloc(&dd, n * 4); hipMalloc(&ds, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<n / 2 / 256, 256>>>(dx, dd, ds, n);
    std::vector<float> rd(n), rs(n);
    hipMemcpy(rd.data(), dd, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rs.data(), ds, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_normal = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&rd[i], &rs[i], 4) != 0) {
            ++bad;
            unsigned xb; memcpy(&xb, &x[i], 4);
            unsigned e = (xb >> 23) & 0xff;
            if (e > 24 && e < 250) { if (bad_normal < 10) printf("x=%a dot=%a sub=%a\n", x[i], rd[i], rs[i]); ++bad_normal; }
        }
    }
    printf("mismatches %ld of %d (with |x| in the normal range away from the ends: %ld)\n", bad, n, bad_normal);
    return 0;
}
