// ilqr_lane.hip -- iLQR.solve (tfmpc/solvers/ilqr.py:214-355) for tiny problems: ONE LANE per
// problem instance, 64 independent solves per wavefront, each lane running its own iteration
// loop (lanes whose instance has converged idle until the wave's slowest instance is done).
// Used by tfmpc_ilqr_solve_f32 for the 2-D navigation envs (BASELINE configs[3]: Navigation,
// n = m = 2, T = 50, batch 16 384) once the batch is large enough to fill lanes.  Same
// equations and quirks as the wave-per-instance path (ilqr_core.h); step-local matrices live
// in registers (small_linalg.h); the nominal trajectory and the gains live in the wave's LDS
// ([slot][lane], bank-conflict free) when the horizon fits (T <= 62), else in HBM slabs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "envs.h"
#include "lqr_kernels.h"
#include "ilqr_trace.h"
#include "options.h"
#include "small_linalg.h"

#include "ilqr_lane_kernels.h"

namespace tfmpc {

// projected_newton_qp (optimization.py:6-101) for two variables, one QP per LANE: the stand-alone entry point
// tfmpc_boxqp_f32 at m = 2 runs the very function the lane kernels call in their backward pass (closed form + iteration)
__global__ __launch_bounds__(64) void boxqp_lane2_kernel(int B, const float *H, const float *q, const float *low,
                                                         const float *high, const float *x0, float *x, float *free_mask,
                                                         int32_t *status)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    Mat<2, 2> Hm;
    float qv[2], lo[2], hi[2], xv[2];
    bool fre[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        Hm(i, 0) = H[(size_t)b * 4 + 2 * i]; Hm(i, 1) = H[(size_t)b * 4 + 2 * i + 1];
        qv[i] = q[(size_t)b * 2 + i]; lo[i] = low[(size_t)b * 2 + i]; hi[i] = high[(size_t)b * 2 + i];
        xv[i] = x0[(size_t)b * 2 + i];
    }
    const int rc = boxqp_lane<2>(Hm, qv, lo, hi, xv, fre);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        x[(size_t)b * 2 + i] = xv[i];
        if (free_mask) free_mask[(size_t)b * 2 + i] = fre[i] ? 1.0f : 0.0f;
    }
    if (status) status[b] = rc;
}

int boxqp_lane2_launch(int B, const float *H, const float *q, const float *low, const float *high, const float *x0,
                       float *x, float *free_mask, int32_t *status, hipStream_t stream)
{
    hipLaunchKernelGGL(boxqp_lane2_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, B, H, q, low, high, x0, x, free_mask, status);
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}

bool ilqr_lane_group_fits(int T) { return GroupStore<2, 2, 4, 3>::bytes(T) <= 64 * 1024; }

// The group kernel must not run from an instantiation that spills (ilqr_lane_kernels.h).  The build is held to that by
// tests/test_lane_group_private_segment_cpu.py; a library built by another compiler is asked here, once per process, and answers with the wave kernel.
template <int KIND>
static bool group_kernels_spill_nothing()
{
    static const bool clean = [] {
        const void *kerns[2] = {reinterpret_cast<const void *>(ilqr_group_solve_kernel<KIND, 2, 2, 1>), reinterpret_cast<const void *>(ilqr_group_solve_kernel<KIND, 2, 2, 4>)};
        for (const void *k : kerns) {
            hipFuncAttributes attr{};
            if (hipFuncGetAttributes(&attr, k) != hipSuccess || attr.localSizeBytes != 0 || attr.numRegs > 256) return false;
        }
        return true;
    }();
    return clean;
}

bool ilqr_lane_supported(const TfmpcEnv &env)
{
    if (env.n != 2 || env.m != 2) return false;
    if (env.kind == TFMPC_ENV_NAVLQR && !group_kernels_spill_nothing<TFMPC_ENV_NAVLQR>()) return false;
    if (env.kind == TFMPC_ENV_NAVIGATION && !group_kernels_spill_nothing<TFMPC_ENV_NAVIGATION>()) return false;
    if (env.kind == TFMPC_ENV_NAVLQR) return true;
    if (env.kind == TFMPC_ENV_NAVIGATION) return env.stride[1] == 0 && env.stride[2] == 0 && env.n_zones <= 8;
    return false;
}

constexpr size_t kQueueBytes = 256;
// Workspace of the group kernel beyond the five slabs every solve kernel gets: one scratch block per wavefront
// (candidate trajectories of the line search, ScratchSink).
size_t ilqr_lane_extra_workspace_bytes(int B, int n, int m, int T)
{
    if (n != 2 || m != 2 || B <= 0) return 0;
    const size_t blocks = B <= kOneGroupMaxBatch ? (size_t)B : ((size_t)B + 3) / 4;
    return blocks * (size_t)ScratchSink<2, 2>::rows(T) * 64 * sizeof(float) + 256 + kQueueBytes;      // + the instance queue's counter
}

// Wavefronts of the group kernel the chip holds at once (what a persistent grid is sized by).  The answer depends on the kernel variant, on
// the DEVICE and on the dynamic LDS of the launch -- which grows with the horizon (~208 T bytes) -- so it is cached per (device, LDS bytes)
// and not per process: a first solve at a long horizon (few blocks per CU) must not size the grid of every later T = 50 launch
// (a call-order-dependent 2-3 x loss).  The two driver queries cost microseconds on a miss.
template <class Kern>
static int resident_blocks(Kern kern, size_t lds)
{
    struct Entry { int dev; size_t lds; int blocks; };
    static std::mutex lock;
    static std::vector<Entry> cache;            // per kernel variant (this function is a template)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    std::lock_guard<std::mutex> g(lock);
    for (const Entry &e : cache)
        if (e.dev == dev && e.lds == lds) return e.blocks;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), 64, lds) != hipSuccess || per_cu < 1) per_cu = 8;
    if (dev < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    const int blocks = per_cu * cus;
    if (dev >= 0) cache.push_back(Entry{dev, lds, blocks});
    return blocks;
}

// grid of the last persistent group launch of this thread (tfmpc_ilqr_last_group_grid: a diagnostic, like tfmpc_ilqr_last_kernel_name)
static thread_local int g_last_group_grid = 0;

template <int KIND>
static int group_launch(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, const SolveArgsLane &a, hipStream_t stream)
{
    constexpr int PRE = LaneEnv<KIND, 2, 2>::kPre;
    const int B = a.B, T = a.T;
    const dim3 block(64);
    const size_t lds1 = GroupStore<2, 2, 1, PRE>::bytes(T), lds4 = GroupStore<2, 2, 4, PRE>::bytes(T);
    if (hipMemsetAsync(a.queue, 0, sizeof(int), stream) != hipSuccess) return TFMPC_ERR_LAUNCH;
    if (B <= kOneGroupMaxBatch) {    // the chip has room for a wavefront per instance: no divergence between groups
        hipLaunchKernelGGL((ilqr_group_solve_kernel<KIND, 2, 2, 1>), dim3(B), block, lds1, stream, env, cfg, a);
    } else {
        // persistent grid: as many wavefronts as are resident at once (or fewer, if the batch is smaller); the groups
        // pull instances from the queue until it is empty
        const int resident = resident_blocks(ilqr_group_solve_kernel<KIND, 2, 2, 4>, lds4);
        const int blocks = std::min((B + 3) / 4, resident);
        g_last_group_grid = blocks;
        hipLaunchKernelGGL((ilqr_group_solve_kernel<KIND, 2, 2, 4>), dim3(blocks), block, lds4, stream, env, cfg, a);
    }
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}

int ilqr_lane_solve_launch(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, int B, int T, const float *x0,
                           const float *u_init, float *states, float *actions, float *costs, int32_t *iterations,
                           int32_t *status, float *wsK, float *wsk, float *wsx, float *wsu, float *wsc, void *extra,
                           const TraceArgs &trace, hipStream_t stream)
{
    SolveArgsLane a{B, T, x0, u_init, states, actions, costs, iterations, status, wsK, wsk, wsx, wsu, wsc};
    a.trace = trace;
    {
        const bool per_lane = option_is(kOptIlqrKernel, "lane1") && !trace.rows;      // (only the group kernel records a trace)
        const size_t glds = GroupStore<2, 2, 4, 3>::bytes(T);
        if (!per_lane && glds <= 64 * 1024) {
            // [queue counter: 256 bytes][scratch blocks]
            a.queue = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(extra) + 255) & ~(uintptr_t)255);
            a.scratch = reinterpret_cast<float *>(reinterpret_cast<char *>(a.queue) + kQueueBytes);
            a.stored = group_stored_candidates(option_int(kOptGroupStored, 0));
            if (env.kind == TFMPC_ENV_NAVLQR) return group_launch<TFMPC_ENV_NAVLQR>(env, cfg, a, stream);
            if (env.kind == TFMPC_ENV_NAVIGATION) return group_launch<TFMPC_ENV_NAVIGATION>(env, cfg, a, stream);
            return TFMPC_ERR_UNSUPPORTED;
        }
    }
    const dim3 grid((B + 63) / 64), block(64);
    const size_t lds = LdsStore<2, 2>::bytes(T);
    const bool use_lds = lds <= kMaxLdsBytes;          // T <= 62 at n = m = 2; else HBM slabs
#define TFMPC_LANE_LAUNCH(KIND_, USE_)                                                                   \
    do {                                                                                                 \
        auto kern = ilqr_lane_solve_kernel<KIND_, 2, 2, USE_>;                                           \
        if (USE_ && lds > 64 * 1024 &&                                                                   \
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)     \
            return TFMPC_ERR_LAUNCH;                                                                     \
        hipLaunchKernelGGL(kern, grid, block, USE_ ? lds : 0, stream, env, cfg, a);                      \
    } while (0)
    if (env.kind == TFMPC_ENV_NAVLQR) {
        if (use_lds) TFMPC_LANE_LAUNCH(TFMPC_ENV_NAVLQR, true); else TFMPC_LANE_LAUNCH(TFMPC_ENV_NAVLQR, false);
    } else if (env.kind == TFMPC_ENV_NAVIGATION) {
        if (use_lds) TFMPC_LANE_LAUNCH(TFMPC_ENV_NAVIGATION, true); else TFMPC_LANE_LAUNCH(TFMPC_ENV_NAVIGATION, false);
    } else {
        return TFMPC_ERR_UNSUPPORTED;
    }
#undef TFMPC_LANE_LAUNCH
    return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH;
}

}  // namespace tfmpc

extern "C" int tfmpc_ilqr_last_group_grid(void) { return tfmpc::g_last_group_grid; }
