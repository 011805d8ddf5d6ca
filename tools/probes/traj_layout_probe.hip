// traj_layout_probe.hip -- does the trajectory LAYOUT bound the cfg5 kernel?  2048 waves x 16 columns stream the
// trajectories of 32 768 instances (n = 32, T = 100) the way ilqr_adjoint_mfma_kernel's rollouts do (per step: two 16-byte
// loads per lane of u_hat_t prefetched 4 steps ahead, a selector byte, stores of x_{t+1} and u_t, a cost), with a
// dependent fma chain standing in for the step's arithmetic.  SCATTER: instance-major arrays (a load instruction touches
// 16 segments of 64 bytes, 12.8 KB apart); else: wave-major (one contiguous KB per instruction).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/traj_layout_probe.hip -o tools/probes/ab/traj_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
#define GL __attribute__((address_space(1)))
template <bool SCATTER, bool STORE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(const float *u, float *xo, float *uo, float *co,
                                                                                      const unsigned char *ks, int T, int passes, int filler, float *sink)
{
    const int lane = threadIdx.x, j = lane & 15, q = lane >> 4, w = blockIdx.x, n = 32;
    const size_t b = (size_t)w * 16 + j;
    float acc = 0.0f;
    for (int p = 0; p < passes; ++p) {
        f32x4 ring[4][2];
        unsigned kb[4];
        auto addr = [&](int t, int tile) -> size_t {
            return SCATTER ? (b * T + t) * n + 16 * tile + 4 * q : (((size_t)w * T + t) * 2 + tile) * 256 + 4 * lane;
        };
        auto request = [&](int t, int d) {
            ring[d][0] = *(const GL f32x4 *)(u + addr(t, 0));
            ring[d][1] = *(const GL f32x4 *)(u + addr(t, 1));
            kb[d] = *(const GL unsigned char *)(ks + (SCATTER ? (b * T + t) * 4 + q : ((size_t)w * T + t) * 64 + lane));
        };
        for (int d = 0; d < 4; ++d) request(d, d);
        for (int t0 = 0; t0 < T; t0 += 4) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int t = t0 + d;
                f32x4 a0 = ring[d][0], a1 = ring[d][1];
                float s = a0.x + a1.y + (float)kb[d];
                if (t + 4 < T) request(t + 4, d);
                for (int i = 0; i < filler; ++i) s = fmaf(s, 1.0000001f, 0.5f);
                acc += s;
                if (STORE) {
                    a0.x = s; a1.x = s;
                    *(GL f32x4 *)(xo + addr(t, 0)) = a0;
                    *(GL f32x4 *)(xo + addr(t, 1)) = a1;
                    *(GL f32x4 *)(uo + addr(t, 0)) = a1;
                    *(GL f32x4 *)(uo + addr(t, 1)) = a0;
                    if (q == 0) *(GL float *)(co + (SCATTER ? b * T + t : ((size_t)w * T + t) * 16 + j)) = s;
                }
            }
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}
int main()
{
    const int B = 32768, T = 100, n = 32, W = B / 16;
    const size_t N = (size_t)B * T * n;
    float *u, *xo, *uo, *co, *sink; unsigned char *ks;
    hipMalloc(&u, N * 4); hipMalloc(&xo, N * 4); hipMalloc(&uo, N * 4); hipMalloc(&co, (size_t)B * T * 4); hipMalloc(&ks, (size_t)B * T * 4); hipMalloc(&sink, 4);
    hipMemset(u, 0, N * 4); hipMemset(ks, 0, (size_t)B * T * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int filler : {0, 100, 300, 600}) for (int store = 0; store < 2; ++store) for (int sc = 0; sc < 2; ++sc) {
        const int passes = store ? 20 : 60;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (sc && store) hipLaunchKernelGGL((probe<true, true>), dim3(W), dim3(64), 0, 0, u, xo, uo, co, ks, T, passes, filler, sink);
            if (sc && !store) hipLaunchKernelGGL((probe<true, false>), dim3(W), dim3(64), 0, 0, u, xo, uo, co, ks, T, passes, filler, sink);
            if (!sc && store) hipLaunchKernelGGL((probe<false, true>), dim3(W), dim3(64), 0, 0, u, xo, uo, co, ks, T, passes, filler, sink);
            if (!sc && !store) hipLaunchKernelGGL((probe<false, false>), dim3(W), dim3(64), 0, 0, u, xo, uo, co, ks, T, passes, filler, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        const double bytes = (double)passes * B * T * (n * 4 + 4 + (store ? 2 * n * 4 + 4 : 0));
        printf("filler %4d %s %s: %7.2f ms for %d passes = %6.1f us per wave-step pair, %5.2f TB/s\n", filler, store ? "load+store" : "load only ",
               sc ? "instance-major" : "wave-major    ", best, passes, best * 1e3 / (passes * T), bytes / best / 1e9);
    }
    return 0;
}
