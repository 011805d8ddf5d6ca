// Where do the waves of a workgroup land?  Every wave records HW_REG_HW_ID (SIMD, CU, SE) for (block, wave); the host prints, per block size, the
// histogram of SIMD ids by wave index and how many waves each SIMD of a CU holds when several blocks share the CU.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/wave_placement_probe.hip -o tools/probes/ab/wave_placement_probe && tools/probes/ab/wave_placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void probe(unsigned *out, int spin)
{
    extern __shared__ float lds[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;          // keep the wave resident while the grid fills the chip
    lds[threadIdx.x] = v;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = id;
        out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (v == 12345.678f) out[0] = 0;
}
int main()
{
    for (int nw : {1, 2, 4, 8}) {
        for (int blocks : {256, 512, 1024, 2048}) {
            if (nw * blocks > 4096 || nw * blocks < 1024) continue;
            const int waves = nw * blocks;
            unsigned *d; hipMalloc(&d, waves * 2 * sizeof(unsigned));
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * nw), 20 * 1024, 0, d, 200000);
            std::vector<unsigned> h(waves * 2); hipMemcpy(h.data(), d, waves * 2 * sizeof(unsigned), hipMemcpyDeviceToHost); hipFree(d);
            int hist[8][4] = {};
            std::map<unsigned, std::vector<int>> per_cu;          // (xcc, se, cu) -> waves per SIMD
            for (int w = 0; w < waves; ++w) {
                const unsigned id = h[2 * w], xcc = h[2 * w + 1] & 0xF;
                const int simd = (id >> 4) & 3, cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
                hist[w % nw][simd]++;
                auto &v = per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu];
                if (v.empty()) v.assign(4, 0);
                v[simd]++;
            }
            int worst = 0, cus = (int)per_cu.size(); double spread = 0;
            for (auto &kv : per_cu) { int mx = 0, mn = 1 << 30; for (int s : kv.second) { mx = s > mx ? s : mx; mn = s < mn ? s : mn; } worst = mx > worst ? mx : worst; spread += mx - mn; }
            printf("waves/block %d, blocks %4d: CUs used %3d, max waves on one SIMD %d, mean (max - min) per CU %.2f | SIMD of wave index:", nw, blocks, cus, worst, spread / cus);
            for (int k = 0; k < nw; ++k) printf(" w%d[%d %d %d %d]", k, hist[k][0], hist[k][1], hist[k][2], hist[k][3]);
            printf("\n");
        }
    }
    return 0;
}
