// lqr_dispatch.hip -- extern "C" entry points of the LQR path (include/tfmpc_hip.h):
// argument checks, kernel-variant choice, launch.  No allocation, no sync.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "lqr_kernels.h"
#include "options.h"

using namespace tfmpc;

namespace {

bool want_mfma(int n, int m)
{
    if (option_is(kOptLqrKernel, "generic")) return false;   // testing / A-B timing
    return lqr_mfma_supported(n, m);
}

// n <= 32, m <= 16 beyond the 16 x 8 tile: one wave per instance, 2 x 2 tiles on the bf16 matrix cores
// (TFMPC_LQR_KERNEL=generic|block select the older kernels)
bool want_mfma32(int n, int m)
{
    if (option_is(kOptLqrKernel, "generic") || option_is(kOptLqrKernel, "block")) return false;
    return lqr_mfma32_supported(n, m);
}

// lane-per-instance pays once there are enough instances to fill lanes; below that the
// wave-per-instance kernel has the shorter critical path.  TFMPC_LQR_KERNEL=lane|generic forces.
constexpr int kLaneMinBatch = 32;

bool want_lane(int n, int m, int B)
{
    if (!lqr_lane_supported(n, m)) return false;
    if (option_is(kOptLqrKernel, "generic")) return false;
    if (option_is(kOptLqrKernel, "lane")) return true;
    return B >= kLaneMinBatch;
}

// Workgroup-per-instance kernel (lqr_block.hip) vs wave-per-instance (lqr_generic.hip), measured on MI355X at T = 50
// (tools/block_vs_wave.py): the block kernel has the shorter critical path at every shape (B = 256: 1.4-3.9x), but it
// holds 3 workgroups per CU, so once the batch fills the chip the wave kernel's many resident waves win below
// n + m ~ 28 (B = 8192: 0.4-0.7x there, 1.5-3.4x above).  TFMPC_LQR_KERNEL=block|generic forces.
constexpr int kBlockFrom = 28;
constexpr int kBlockMaxSmallBatch = 2048;

bool want_block(int n, int m, int B)
{
    if (lqr_block_smem_bytes(n, m) > kMaxLdsBytes) return false;
    if (option_is(kOptLqrKernel, "generic")) return false;
    if (option_is(kOptLqrKernel, "block")) return true;
    if (lqr_lane_supported(n, m)) return false;           // tiny shapes: one wave (or one lane) is already the short path
    return n + m >= kBlockFrom || B <= kBlockMaxSmallBatch;
}

int check_common(int B, int n, int m, int T, const void *F, const void *f, const void *C, const void *c, bool general)
{
    if (B < 0 || n <= 0 || m <= 0 || T < 0) return TFMPC_ERR_ARG;
    if (!F || !f || !C || !c) return TFMPC_ERR_ARG;
    if ((general || !want_mfma(n, m)) && lqr_generic_smem_bytes(n, m) > kMaxLdsBytes) return TFMPC_ERR_UNSUPPORTED;
    return TFMPC_OK;
}

int run(const LqrArgs &a, bool bw, bool fw, bool general, void *stream)
{
    if (a.B == 0) return TFMPC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // `general`: C is not symmetric.  Only the wave kernel restates lqr.py:74-105 term by term (Q_ux and Q_xu
    // read separately, pivoted general inverse, four-term update, no symmetrisation); every other variant uses
    // the symmetry of C and V.
    if (general) return lqr_generic_launch(a, bw, fw, s);
    if (want_mfma(a.n, a.m)) return lqr_mfma_launch(a, bw, fw, s);
    if (want_mfma32(a.n, a.m)) return lqr_mfma32_launch(a, bw, fw, s);
    // 16-bit policy / value outputs (tfmpc_lqr_*_bf16out_f32) are written by the two kernels above and the wave kernel
    if (a.K16 || a.k16 || a.V16 || a.v16 || a.cst16)
        return lqr_generic_smem_bytes(a.n, a.m) <= kMaxLdsBytes ? lqr_generic_launch(a, bw, fw, s) : TFMPC_ERR_UNSUPPORTED;
    if (want_lane(a.n, a.m, a.B)) return lqr_lane_launch(a, bw, fw, s);
    if (want_block(a.n, a.m, a.B)) return lqr_block_launch(a, bw, fw, s);
    return lqr_generic_launch(a, bw, fw, s);
}

}  // namespace

extern "C" {

int tfmpc_version(void) { return 310; }      // 310: tfmpc_ilqr_workspace_bytes_for, the costate kernel's coefficient slab behind the groups' slices; 300 (round 6): tfmpc_ilqr_solve_trace_qp_f32, TFMPC_ILQR_LQ_REUSE; 200: TFMPC_TRACE_COLS 11, TfmpcEnv::coupling_shift, status bits 0x20 / 0x40

const char *tfmpc_lqr_kernel_name(int n, int m, int T)
{
    (void)T;
    if (n <= 0 || m <= 0) return "invalid";
    if (want_mfma(n, m)) return (n == 16 && m == 8) ? "mfma_16x8" : "mfma_16x8 (zero-padded)";
    if (want_mfma32(n, m)) return (n == 32 && m == 16) ? "mfma_32x16" : "mfma_32x16 (zero-padded)";
    if (lqr_lane_supported(n, m)) return "lane (batch >= 32) / generic_wave";
    if (lqr_generic_smem_bytes(n, m) > kMaxLdsBytes) return "unsupported";
    if (want_block(n, m, 1 << 30)) return "block_mfma_f32";
    if (want_block(n, m, 1)) return "block_mfma_f32 (batch <= 2048) / generic_wave";
    return "generic_wave";
}

size_t tfmpc_lqr_workspace_bytes(int B, int n, int m, int T)
{
    if (B <= 0 || n <= 0 || m <= 0 || T <= 0) return 0;
    return (size_t)B * T * m * (n + 1) * sizeof(float);
}


}  // extern "C"

namespace {

struct Out16 { uint16_t *K, *k, *V, *v, *cst; };

int backward_impl(bool general, int B, int n, int m, int T, const float *F, long strideF, const float *f,
                  long stride_f, const float *C, long strideC, const float *c, long stride_c,
                  float *K, float *k, float *V, float *v, float *cst, int32_t *status, void *stream,
                  const Out16 &o16 = Out16{})
{
    int rc = check_common(B, n, m, T, F, f, C, c, general);
    if (rc != TFMPC_OK) return rc;
    if (T > 0 && B > 0 && (!K || !k)) return TFMPC_ERR_ARG;
    LqrArgs a{};
    a.B = B; a.n = n; a.m = m; a.T = T;
    a.F = F; a.f = f; a.C = C; a.c = c;
    a.sF = strideF; a.sf = stride_f; a.sC = strideC; a.sc = stride_c;
    a.K = K; a.k = k; a.sK = (long)T * m * n; a.sk = (long)T * m;
    a.V = V; a.v = v; a.cst = cst; a.status = status;
    a.K16 = o16.K; a.k16 = o16.k; a.V16 = o16.V; a.v16 = o16.v; a.cst16 = o16.cst;
    return run(a, true, false, general, stream);
}

int forward_impl(bool general, int B, int n, int m, int T, const float *F, long strideF, const float *f,
                 long stride_f, const float *C, long strideC, const float *c, long stride_c,
                 const float *K, long strideK, const float *k, long stride_k, const float *x0,
                 float *states, float *actions, float *costs, void *stream)
{
    int rc = check_common(B, n, m, T, F, f, C, c, general);
    if (rc != TFMPC_OK) return rc;
    if (B > 0 && (!x0 || !states || !costs)) return TFMPC_ERR_ARG;
    if (B > 0 && T > 0 && (!K || !k || !actions)) return TFMPC_ERR_ARG;
    LqrArgs a{};
    a.B = B; a.n = n; a.m = m; a.T = T;
    a.F = F; a.f = f; a.C = C; a.c = c; a.x0 = x0;
    a.sF = strideF; a.sf = stride_f; a.sC = strideC; a.sc = stride_c;
    a.K = const_cast<float *>(K); a.k = const_cast<float *>(k); a.sK = strideK; a.sk = stride_k;
    a.states = states; a.actions = actions; a.costs = costs;
    return run(a, false, true, general, stream);
}

int solve_impl(bool general, int B, int n, int m, int T, const float *F, long strideF, const float *f,
               long stride_f, const float *C, long strideC, const float *c, long stride_c,
               const float *x0, float *states, float *actions, float *costs, float *K, float *k,
               float *V, float *v, float *cst, int32_t *status, void *workspace,
               size_t workspace_bytes, void *stream, const Out16 &o16 = Out16{})
{
    int rc = check_common(B, n, m, T, F, f, C, c, general);
    if (rc != TFMPC_OK) return rc;
    if (B > 0 && (!x0 || !states || !costs)) return TFMPC_ERR_ARG;
    if (B > 0 && T > 0 && !actions) return TFMPC_ERR_ARG;
    if (B > 0 && T > 0 && (!K || !k)) {
        // gains are not a requested output: keep them in caller-provided scratch
        if (!workspace || workspace_bytes < tfmpc_lqr_workspace_bytes(B, n, m, T)) return TFMPC_ERR_WORKSPACE;
        float *w = static_cast<float *>(workspace);
        if (!K) { K = w; }
        if (!k) { k = w + (size_t)B * T * m * n; }
    }
    LqrArgs a{};
    a.B = B; a.n = n; a.m = m; a.T = T;
    a.F = F; a.f = f; a.C = C; a.c = c; a.x0 = x0;
    a.sF = strideF; a.sf = stride_f; a.sC = strideC; a.sc = stride_c;
    a.K = K; a.k = k; a.sK = (long)T * m * n; a.sk = (long)T * m;
    a.V = V; a.v = v; a.cst = cst;
    a.states = states; a.actions = actions; a.costs = costs; a.status = status;
    a.K16 = o16.K; a.k16 = o16.k; a.V16 = o16.V; a.v16 = o16.v; a.cst16 = o16.cst;
    return run(a, true, true, general, stream);
}

}  // namespace

extern "C" {

#define TFMPC_LQR_BACKWARD_PARAMS                                                                              \
    int B, int n, int m, int T, const float *F, long strideF, const float *f, long stride_f, const float *C,   \
        long strideC, const float *c, long stride_c, float *K, float *k, float *V, float *v, float *cst,       \
        int32_t *status, void *stream
#define TFMPC_LQR_BACKWARD_ARGS B, n, m, T, F, strideF, f, stride_f, C, strideC, c, stride_c, K, k, V, v, cst, status, stream
#define TFMPC_LQR_FORWARD_PARAMS                                                                               \
    int B, int n, int m, int T, const float *F, long strideF, const float *f, long stride_f, const float *C,   \
        long strideC, const float *c, long stride_c, const float *K, long strideK, const float *k,             \
        long stride_k, const float *x0, float *states, float *actions, float *costs, void *stream
#define TFMPC_LQR_FORWARD_ARGS \
    B, n, m, T, F, strideF, f, stride_f, C, strideC, c, stride_c, K, strideK, k, stride_k, x0, states, actions, costs, stream
#define TFMPC_LQR_SOLVE_PARAMS                                                                                 \
    int B, int n, int m, int T, const float *F, long strideF, const float *f, long stride_f, const float *C,   \
        long strideC, const float *c, long stride_c, const float *x0, float *states, float *actions,           \
        float *costs, float *K, float *k, float *V, float *v, float *cst, int32_t *status, void *workspace,    \
        size_t workspace_bytes, void *stream
#define TFMPC_LQR_SOLVE_ARGS                                                                                   \
    B, n, m, T, F, strideF, f, stride_f, C, strideC, c, stride_c, x0, states, actions, costs, K, k, V, v, cst, \
        status, workspace, workspace_bytes, stream

int tfmpc_lqr_backward_f32(TFMPC_LQR_BACKWARD_PARAMS) { return backward_impl(false, TFMPC_LQR_BACKWARD_ARGS); }
int tfmpc_lqr_forward_f32(TFMPC_LQR_FORWARD_PARAMS) { return forward_impl(false, TFMPC_LQR_FORWARD_ARGS); }
int tfmpc_lqr_solve_f32(TFMPC_LQR_SOLVE_PARAMS) { return solve_impl(false, TFMPC_LQR_SOLVE_ARGS); }
// C not symmetric: the reference's term-by-term recursion (see tfmpc_hip.h)
int tfmpc_lqr_backward_general_f32(TFMPC_LQR_BACKWARD_PARAMS) { return backward_impl(true, TFMPC_LQR_BACKWARD_ARGS); }
int tfmpc_lqr_forward_general_f32(TFMPC_LQR_FORWARD_PARAMS) { return forward_impl(true, TFMPC_LQR_FORWARD_ARGS); }
int tfmpc_lqr_solve_general_f32(TFMPC_LQR_SOLVE_PARAMS) { return solve_impl(true, TFMPC_LQR_SOLVE_ARGS); }

// Policy / value-function outputs in 16-bit containers (lqr.py:107-129 is what they hold)
int tfmpc_lqr_solve_bf16out_f32(int B, int n, int m, int T, const float *F, long strideF, const float *f, long stride_f,
                                const float *C, long strideC, const float *c, long stride_c, const float *x0,
                                float *states, float *actions, float *costs, uint16_t *K16, uint16_t *k16, uint16_t *V16,
                                uint16_t *v16, uint16_t *cst16, int32_t *status, void *workspace, size_t workspace_bytes,
                                void *stream)
{
    return solve_impl(false, B, n, m, T, F, strideF, f, stride_f, C, strideC, c, stride_c, x0, states, actions, costs,
                      nullptr, nullptr, nullptr, nullptr, nullptr, status, workspace, workspace_bytes, stream,
                      Out16{K16, k16, V16, v16, cst16});
}

int tfmpc_lqr_backward_bf16out_f32(int B, int n, int m, int T, const float *F, long strideF, const float *f, long stride_f,
                                   const float *C, long strideC, const float *c, long stride_c, uint16_t *K16,
                                   uint16_t *k16, uint16_t *V16, uint16_t *v16, uint16_t *cst16, int32_t *status,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    // the sweep still needs its fp32 gains (K_t feeds V_t): they live in the caller's scratch
    if (B > 0 && T > 0 && (!workspace || workspace_bytes < tfmpc_lqr_workspace_bytes(B, n, m, T))) return TFMPC_ERR_WORKSPACE;
    float *w = static_cast<float *>(workspace);
    return backward_impl(false, B, n, m, T, F, strideF, f, stride_f, C, strideC, c, stride_c, w,
                         w ? w + (size_t)B * T * m * n : nullptr, nullptr, nullptr, nullptr, status, stream,
                         Out16{K16, k16, V16, v16, cst16});
}


}  // extern "C"
