// ilqr_kernels.hip -- gfx950 kernels and extern "C" entry points of the iLQR path
// (include/tfmpc_hip.h): rollout (iLQR.start), derivatives, backward, forward, the
// fused whole-solve kernel and the stand-alone box-QP.  One wavefront per problem
// instance throughout; see ilqr_core.h for the per-step arithmetic.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "ilqr_adjoint.h"
#include "ilqr_core.h"
#include "ilqr_lq_mfma.h"
#include "ilqr_trace.h"
#include "ilqr_wave_kernels.h"
#include "lqr_kernels.h"
#include "options.h"

namespace tfmpc {

// lane-per-instance fused solve for tiny envs (ilqr_lane.hip)
bool ilqr_lane_supported(const TfmpcEnv &env);
int ilqr_lane_solve_launch(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, int B, int T, const float *x0,
                           const float *u_init, float *states, float *actions, float *costs, int32_t *iterations,
                           int32_t *status, float *wsK, float *wsk, float *wsx, float *wsu, float *wsc, void *extra,
                           const TraceArgs &trace, hipStream_t stream);
size_t ilqr_lane_extra_workspace_bytes(int B, int n, int m, int T);
bool ilqr_lane_group_fits(int T);
int boxqp_lane2_launch(int B, const float *H, const float *q, const float *low, const float *high, const float *x0,
                       float *x, float *free_mask, int32_t *status, hipStream_t stream);   // m = 2: one QP per lane       // the 16-lanes-per-instance kernel (the one that records a decision trace) serves this horizon


__global__ __launch_bounds__(kWave) void boxqp_kernel(int m, const float *H, const float *q, const float *low,
                                                      const float *high, const float *x0, float *x, float *free_mask,
                                                      int32_t *status)
{
    extern __shared__ float smem[];
    const int b = blockIdx.x, lane = lane_id();
    IlqrSmem s;
    float *p = ilqr_carve(s, smem, 1, m);
    const int ldm = s.ldm;
    float *Hm = p, *qv = p + m * ldm;
    load_matrix(Hm, ldm, H + (size_t)b * m * m, m, m);
    for (int i = lane; i < m; i += kWave) {
        qv[i] = q[(size_t)b * m + i];
        s.qlo[i] = low[(size_t)b * m + i];
        s.qhi[i] = high[(size_t)b * m + i];
        s.qx[i] = x0[(size_t)b * m + i];
    }
    wsync();
    const int rc = boxqp(s, Hm, ldm, qv);
    wsync();
    for (int i = lane; i < m; i += kWave) {
        x[(size_t)b * m + i] = s.qx[i];
        if (free_mask) free_mask[(size_t)b * m + i] = s.qfree[i];
    }
    if (status && lane == 0) status[b] = rc;
}

// ---- host side ---------------------------------------------------------------------
static size_t ilqr_smem_bytes(int kind, int n, int m, int zones)
{
    return (ilqr_smem_floats(n, m) + env_lds_floats(kind, n, m, zones) + 2 * (size_t)m) * sizeof(float);
}

// the fused solve of HVAC / Reservoir uses the adjoint backward pass and a reduced LDS slice
static size_t ilqr_solve_smem_bytes(int kind, int n, int m, int zones)
{
    const bool adjoint = kind == TFMPC_ENV_HVAC || kind == TFMPC_ENV_RESERVOIR;
    const size_t core = adjoint ? ilqr_adjoint_smem_floats(n, m) : ilqr_smem_floats(n, m);
    return (core + env_lds_floats(kind, n, m, zones) + 2 * (size_t)m) * sizeof(float);
}

static int check_env(const TfmpcEnv *env)
{
    if (!env || env->n <= 0 || env->m <= 0 || !env->low || !env->high) return TFMPC_ERR_ARG;
    if (env->kind < TFMPC_ENV_LQ || env->kind > TFMPC_ENV_RESERVOIR) return TFMPC_ERR_ARG;
    if (env->kind != TFMPC_ENV_LQ && env->n != env->m) return TFMPC_ERR_ARG;   // reference envs: action_size == state_size
    if (env->kind == TFMPC_ENV_NAVIGATION && (env->n > 8 || env->n_zones < 0)) return TFMPC_ERR_ARG;
    for (int i = 0; i < TFMPC_ENV_MAX_PARAMS; ++i)
        if (env_param_len(env->kind, i, env->n, env->m, env->n_zones) > 0 && !env->p[i]) return TFMPC_ERR_ARG;
    if (ilqr_smem_bytes(env->kind, env->n, env->m, env->n_zones) > kMaxLdsBytes) return TFMPC_ERR_UNSUPPORTED;
    return TFMPC_OK;
}

template <class Kern>
static int prep(Kern kern, size_t smem)
{
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return TFMPC_ERR_LAUNCH;
    return TFMPC_OK;
}

static int launched() { return hipGetLastError() == hipSuccess ? TFMPC_OK : TFMPC_ERR_LAUNCH; }

#define TFMPC_DISPATCH_KIND(kind, CALL)                                           \
    switch (kind) {                                                               \
    case TFMPC_ENV_LQ: { constexpr int KIND = TFMPC_ENV_LQ; CALL; } break;         \
    case TFMPC_ENV_NAVLQR: { constexpr int KIND = TFMPC_ENV_NAVLQR; CALL; } break; \
    case TFMPC_ENV_NAVIGATION: { constexpr int KIND = TFMPC_ENV_NAVIGATION; CALL; } break; \
    case TFMPC_ENV_HVAC: { constexpr int KIND = TFMPC_ENV_HVAC; CALL; } break;     \
    case TFMPC_ENV_RESERVOIR: { constexpr int KIND = TFMPC_ENV_RESERVOIR; CALL; } break; \
    default: return TFMPC_ERR_ARG;                                                \
    }

}  // namespace tfmpc

using namespace tfmpc;

extern "C" {

int tfmpc_ilqr_rollout_f32(const TfmpcEnv *env, int B, int T, const float *x0, const float *actions, float *states,
                           float *costs, void *stream)
{
    int rc = check_env(env);
    if (rc != TFMPC_OK) return rc;
    if (B < 0 || T < 0) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!x0 || !states || !costs || (T > 0 && !actions)) return TFMPC_ERR_ARG;
    const size_t smem = ilqr_smem_bytes(env->kind, env->n, env->m, env->n_zones);
    hipStream_t st = static_cast<hipStream_t>(stream);
    TFMPC_DISPATCH_KIND(env->kind, {
        auto kern = ilqr_rollout_kernel<KIND>;
        if ((rc = prep(kern, smem)) != TFMPC_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWave), smem, st, *env, T, x0, actions, states, costs);
    });
    return launched();
}

int tfmpc_ilqr_derivatives_f32(const TfmpcEnv *env, int B, int T, const float *states, const float *actions, float *f,
                               float *f_x, float *f_u, float *l, float *l_x, float *l_u, float *l_xx, float *l_uu,
                               float *l_ux, float *l_xu, float *fl, float *fl_x, float *fl_xx, void *stream)
{
    int rc = check_env(env);
    if (rc != TFMPC_OK) return rc;
    if (B < 0 || T < 0) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!states || (T > 0 && !actions)) return TFMPC_ERR_ARG;
    const size_t smem = ilqr_smem_bytes(env->kind, env->n, env->m, env->n_zones);
    hipStream_t st = static_cast<hipStream_t>(stream);
    DerivOut o{f, f_x, f_u, l, l_x, l_u, l_xx, l_uu, l_ux, l_xu, fl, fl_x, fl_xx};
    TFMPC_DISPATCH_KIND(env->kind, {
        auto kern = ilqr_derivatives_kernel<KIND>;
        if ((rc = prep(kern, smem)) != TFMPC_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWave), smem, st, *env, T, states, actions, o);
    });
    return launched();
}

int tfmpc_ilqr_backward_f32(int B, int n, int m, int T, const float *actions, const float *f_x, const float *f_u,
                            const float *l, const float *l_x, const float *l_u, const float *l_xx, const float *l_uu,
                            const float *l_xu, const float *fl, const float *fl_x, const float *fl_xx,
                            const float *low, const float *high, int bounded, const float *mu, long mu_stride,
                            float *K, float *k, float *J, float *dV1, float *dV2, int32_t *status, void *stream)
{
    if (B < 0 || n <= 0 || m <= 0 || T < 0) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!fl || !fl_x || !fl_xx || !low || !high || !mu || !J || !dV1 || !dV2) return TFMPC_ERR_ARG;
    if (T > 0 && (!actions || !f_x || !f_u || !l || !l_x || !l_u || !l_xx || !l_uu || !l_xu || !K || !k)) return TFMPC_ERR_ARG;
    const size_t smem = (ilqr_smem_floats(n, m) + 2 * (size_t)m) * sizeof(float);
    if (smem > kMaxLdsBytes) return TFMPC_ERR_UNSUPPORTED;
    const bool blk = n >= kBlockedFrom;
    int rc = blk ? prep(ilqr_backward_kernel<true>, smem) : prep(ilqr_backward_kernel<false>, smem);
    if (rc != TFMPC_OK) return rc;
    BackwardArgs a{n, m, T, bounded, actions, f_x, f_u, l, l_x, l_u, l_xx, l_uu, l_xu, fl, fl_x, fl_xx, low, high, mu,
                   mu_stride, K, k, J, dV1, dV2, status};
    if (blk) hipLaunchKernelGGL(ilqr_backward_kernel<true>, dim3(B), dim3(kWave), smem, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(ilqr_backward_kernel<false>, dim3(B), dim3(kWave), smem, static_cast<hipStream_t>(stream), a);
    return launched();
}

int tfmpc_ilqr_forward_f32(const TfmpcEnv *env, int B, int T, const float *x, const float *u, const float *K,
                           const float *k, const float *alpha, long alpha_stride, float *states, float *actions,
                           float *costs, float *J, float *residual, void *stream)
{
    int rc = check_env(env);
    if (rc != TFMPC_OK) return rc;
    if (B < 0 || T < 0) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!x || !alpha || !states || !costs || !J || !residual) return TFMPC_ERR_ARG;
    if (T > 0 && (!u || !K || !k || !actions)) return TFMPC_ERR_ARG;
    const size_t smem = ilqr_smem_bytes(env->kind, env->n, env->m, env->n_zones);
    hipStream_t st = static_cast<hipStream_t>(stream);
    TFMPC_DISPATCH_KIND(env->kind, {
        auto kern = ilqr_forward_kernel<KIND>;
        if ((rc = prep(kern, smem)) != TFMPC_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWave), smem, st, *env, T, x, u, K, k, alpha, alpha_stride, states,
                           actions, costs, J, residual);
    });
    return launched();
}

// The ONE description of the solve workspace (tfmpc_ilqr_workspace_bytes sizes it, tfmpc_ilqr_solve_trace_f32 carves it):
// five per-instance slabs -- gains K[T][m][n], k[T][m] and a candidate trajectory x[T+1][n], u[T][m], c[T+1] -- then, each
// from a 256-byte boundary, the line-search scratch of the 2-D lane-group kernel and the wave-major trajectory buffers of
// the 16-instances-per-wave HVAC / Reservoir kernel.  Offsets in bytes from a 256-byte aligned base.
struct IlqrWsLayout {
    size_t K, k, x, u, c, lane, wave, minv, total;
    // kind < 0: the size that serves EVERY env of the shape (tfmpc_ilqr_workspace_bytes); a kind: only the parts the kernels of that env kind
    // read (tfmpc_ilqr_workspace_bytes_for -- the three optional parts are hundreds of MB at large batches: an HVAC batch does not need the LQ
    // kernels' slab, an LQ batch not the costate kernel's buffers).  The offsets of the parts that exist do not depend on it beyond what precedes them.
    IlqrWsLayout(int B, int n, int m, int T, int kind = -1)
    {
        const bool all = kind < 0;
        const bool want_lane = all || kind == TFMPC_ENV_NAVLQR || kind == TFMPC_ENV_NAVIGATION;
        const bool want_wave = all || kind == TFMPC_ENV_HVAC || kind == TFMPC_ENV_RESERVOIR;
        const bool want_minv = all || kind == TFMPC_ENV_LQ;
        const size_t f = sizeof(float);
        K = 0;
        k = K + (size_t)B * T * m * n * f;
        x = k + (size_t)B * T * m * f;
        u = x + (size_t)B * (T + 1) * n * f;
        c = u + (size_t)B * T * m * f;
        lane = round256(c + (size_t)B * (T + 1) * f);
        wave = lane + (want_lane ? round256(ilqr_lane_extra_workspace_bytes(B, n, m, T)) : 0);
        minv = round256(wave + (want_wave ? ilqr_adjoint_mfma_workspace_bytes(B, n, m, T) : 0));
        total = minv + (want_minv ? ilqr_lq_mfma_reuse_workspace_bytes(B, n, m, T)        // -Q_uu^-1 of the LQ env's first pass (ilqr_lq_mfma.hip, REUSE) ...
                                        + ilqr_lq_mfma32_reuse_workspace_bytes(B, n, m, T)     // ... or of its large-tile twin (the two shape ranges are disjoint)
                                  : 0);
    }
    static size_t round256(size_t v) { return (v + 255) & ~(size_t)255; }
};

size_t tfmpc_ilqr_workspace_bytes(int B, int n, int m, int T)
{
    if (B <= 0 || n <= 0 || m <= 0 || T < 0) return 0;
    return IlqrWsLayout(B, n, m, T).total;
}

size_t tfmpc_ilqr_workspace_bytes_for(const TfmpcEnv *env, int B, int T)
{
    if (!env || B <= 0 || env->n <= 0 || env->m <= 0 || T < 0) return 0;
    return IlqrWsLayout(B, env->n, env->m, T, env->kind).total;
}

// which kernel family the last tfmpc_ilqr_solve[_trace]_f32 of this thread went to (a traced solve of HVAC / Reservoir, or of a
// tiny 2-D env forced onto the per-lane kernel, takes a kernel that records a trace: the caller can see which one solved it)
static thread_local const char *g_last_ilqr_kernel = "";

const char *tfmpc_ilqr_last_kernel_name(void) { return g_last_ilqr_kernel; }

int tfmpc_ilqr_solve_f32(const TfmpcEnv *env, const TfmpcIlqrConfig *cfg, int B, int T, const float *x0,
                         const float *u_init, float *states, float *actions, float *costs, int32_t *iterations,
                         int32_t *status, void *workspace, size_t workspace_bytes, void *stream)
{
    return tfmpc_ilqr_solve_trace_f32(env, cfg, B, T, x0, u_init, states, actions, costs, iterations, status, nullptr, 0,
                                      nullptr, workspace, workspace_bytes, stream);
}

int tfmpc_ilqr_solve_trace_f32(const TfmpcEnv *env, const TfmpcIlqrConfig *cfg, int B, int T, const float *x0,
                               const float *u_init, float *states, float *actions, float *costs, int32_t *iterations,
                               int32_t *status, float *trace, int trace_rows, int32_t *trace_len, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    return tfmpc_ilqr_solve_trace_qp_f32(env, cfg, B, T, x0, u_init, states, actions, costs, iterations, status, trace, trace_rows,
                                         trace_len, nullptr, nullptr, workspace, workspace_bytes, stream);
}

int tfmpc_ilqr_solve_trace_qp_f32(const TfmpcEnv *env, const TfmpcIlqrConfig *cfg, int B, int T, const float *x0,
                                  const float *u_init, float *states, float *actions, float *costs, int32_t *iterations,
                                  int32_t *status, float *trace, int trace_rows, int32_t *trace_len, uint8_t *clamp_mask,
                                  uint8_t *qp_iterations, void *workspace, size_t workspace_bytes, void *stream)
{
    if ((clamp_mask != nullptr) != (qp_iterations != nullptr) || (clamp_mask && !trace)) return TFMPC_ERR_ARG;
    int rc = check_env(env);
    if (rc != TFMPC_OK) return rc;
    if (!cfg || B < 0 || T < 0 || cfg->n_alphas < 1 || cfg->n_alphas > TFMPC_MAX_ALPHAS || cfg->max_iterations < 1)
        return TFMPC_ERR_ARG;
    if (trace && (trace_rows < 1 || !trace_len)) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!x0 || !states || !costs || !iterations || !status || (T > 0 && (!u_init || !actions))) return TFMPC_ERR_ARG;
    const int n = env->n, m = env->m;
    const IlqrWsLayout lay(B, n, m, T, env->kind);      // (what THIS env's kernels read: a caller may size by tfmpc_ilqr_workspace_bytes_for or by the shape)
    if (!workspace || workspace_bytes < lay.total || (reinterpret_cast<uintptr_t>(workspace) & 255u)) return TFMPC_ERR_WORKSPACE;
    char *const base = static_cast<char *>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const TraceArgs tr{trace, trace_len, trace ? trace_rows : 0, clamp_mask, qp_iterations};
    const bool traced = trace != nullptr;
    if (traced && hipMemsetAsync(trace_len, 0, (size_t)B * sizeof(int32_t), st) != hipSuccess) return TFMPC_ERR_LAUNCH;
    SolveArgs a{};
    a.B = B; a.T = T; a.x0 = x0; a.u_init = u_init;
    a.states = states; a.actions = actions; a.costs = costs; a.iterations = iterations; a.status = status;
    a.wsK = reinterpret_cast<float *>(base + lay.K);
    a.wsk = reinterpret_cast<float *>(base + lay.k);
    a.wsx = reinterpret_cast<float *>(base + lay.x);
    a.wsu = reinterpret_cast<float *>(base + lay.u);
    a.wsc = reinterpret_cast<float *>(base + lay.c);
    a.trace = tr;
    char *const after_slabs = base + lay.lane;
    char *const after_lane = base + lay.wave;
    {
        // tiny 2-D envs: 16 lanes per instance with a speculative parallel line search (ilqr_lane.hip) at EVERY batch
        // size -- also for one instance it has the shorter critical path (Navigation, T = 50, B = 1: 2.3 ms against
        // 7.2 ms for the wave kernel; tools/small_batch_lane_vs_wave.py).  TFMPC_ILQR_KERNEL=lane|lane1|wave forces.
        const bool forced_wave = option_is(kOptIlqrKernel, "wave");
        if (ilqr_lane_supported(*env) && !forced_wave && !cfg->storage_bf16 && (!traced || ilqr_lane_group_fits(T))) {
            g_last_ilqr_kernel = (option_is(kOptIlqrKernel, "lane1") && !traced) ? "lane (one lane per instance)" : "lane_group (16 lanes per instance, instance queue)";
            return ilqr_lane_solve_launch(*env, *cfg, B, T, x0, u_init, states, actions, costs, iterations, status,
                                          a.wsK, a.wsk, a.wsx, a.wsu, a.wsc,
                                          after_slabs, tr, st);
        }
    }
    {
        // LQ env on the matrix cores (ilqr_lq_mfma.hip); instances it cannot finish (mu > 0 needed)
        // come back flagged and are re-solved from scratch by the wave kernel right behind it
        // (a traced solve stays on these kernels: each records the same row per pass as the wave kernel, and the
        // second-chance launch rewrites the rows of the instances it re-solves)
        const bool forced_wave = option_is(kOptIlqrKernel, "wave");
        if (!forced_wave && !cfg->storage_bf16 && ilqr_lq_box_mfma_supported(*env, T)) {
            // control-limited LQ problems: box-QP in registers, the complete solve loop in one kernel
            IlqrLqArgs la{*env, *cfg, B, T, x0, u_init, states, actions, costs, iterations, status, a.wsK, a.wsk, a.wsu};
            la.trace = tr;
            la.board = a.wsx;                           // (the candidate-trajectory slab: unused by this kernel)
            la.board_bytes = (size_t)B * (T + 1) * n * sizeof(float);
            g_last_ilqr_kernel = "lq_box_mfma (matrix cores, control-limited)";
            return ilqr_lq_box_mfma_launch(la, st);
        }
        if (!forced_wave && !cfg->storage_bf16 && ilqr_lq_mfma32_supported(*env, T)) {
            // BASELINE configs[4]'s literal dims (n <= 32, m <= 16): 2 x 2 tiles, trajectories in HBM; second chance as below
            IlqrLqArgs la{*env, *cfg, B, T, x0, u_init, states, actions, costs, iterations, status, a.wsK, a.wsk, a.wsu,
                          a.wsx, a.wsu, a.wsc};
            la.trace = tr;
            la.wsMinv = lay.total > lay.minv ? reinterpret_cast<float *>(base + lay.minv) : nullptr;
            if ((rc = ilqr_lq_mfma32_launch(la, st)) != TFMPC_OK) return rc;
            a.only_flagged = 1;
            g_last_ilqr_kernel = "lq_mfma32 (matrix cores, 2 x 2 tiles) + wave kernel for flagged instances";
        }
        if (!forced_wave && !cfg->storage_bf16 && ilqr_lq_mfma_supported(*env, T)) {
            IlqrLqArgs la{*env, *cfg, B, T, x0, u_init, states, actions, costs, iterations, status, a.wsK, a.wsk, a.wsu};
            la.trace = tr;
            la.wsMinv = lay.total > lay.minv ? reinterpret_cast<float *>(base + lay.minv) : nullptr;
            if ((rc = ilqr_lq_mfma_launch(la, st)) != TFMPC_OK) return rc;
            a.only_flagged = 1;
            g_last_ilqr_kernel = "lq_mfma (matrix cores) + wave kernel for flagged instances";
        }
    }
    {
        // HVAC / Reservoir at n <= 32: register-resident costate kernel (ilqr_adjoint.hip),
        // bit-identical to the generic wave kernel below (TFMPC_ILQR_KERNEL=wave selects that one)
        const bool forced_wave = option_is(kOptIlqrKernel, "wave");
        const AdjointSolveArgs aa{B, T, x0, u_init, states, actions, costs, iterations, status, a.wsk, a.wsx, a.wsu, a.wsc, after_lane, tr,
                                  option_is(kOptCostateCoupling, "dense") ? 1 : 0};
        // a batch that shares one env: 16 instances per wave with the coupling-matrix products on the matrix
        // cores (ilqr_adjoint_mfma.hip); TFMPC_ILQR_KERNEL=costate_mfma forces it, lean / lean1 the kernels above.
        // storage_bf16: that kernel keeps its trajectories in REAL 16-bit containers (the wave kernel below emulates
        // the format in fp32 containers and serves every env the 16-per-wave kernel does not)
        const bool forced_lean = option_is(kOptIlqrKernel, "lean") || option_is(kOptIlqrKernel, "lean1");
        const bool forced_mfma = option_is(kOptIlqrKernel, "costate_mfma");
        if (!forced_wave && !forced_lean && ilqr_adjoint_mfma_supported(*env, *cfg) &&
            (forced_mfma || cfg->storage_bf16 || traced ||
             B >= ((n > 16 && env->kind != TFMPC_ENV_RESERVOIR) ? kCostateMfmaMinBatchLarge : kCostateMfmaMinBatchSmall)))
        {
            g_last_ilqr_kernel = "costate_mfma (16 instances per wave)";
            return ilqr_adjoint_mfma_launch(*env, *cfg, aa, st);
        }
        if (!forced_wave && !traced && ilqr_adjoint_supported(*env, *cfg)) {
            g_last_ilqr_kernel = "costate (register-resident, one instance per wave)";
            return ilqr_adjoint_launch(*env, *cfg, aa, st);
        }
    }
    const size_t smem = ilqr_solve_smem_bytes(env->kind, n, m, env->n_zones);
    if (!a.only_flagged) g_last_ilqr_kernel = "wave (one instance per wave, any env)";
    if (env->kind == TFMPC_ENV_LQ && n >= kBlockedFrom) {      // the only dense env that comes in large shapes
        auto kern = ilqr_solve_kernel<TFMPC_ENV_LQ, true>;
        if ((rc = prep(kern, smem)) != TFMPC_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWave), smem, st, *env, *cfg, a);
        return launched();
    }
    TFMPC_DISPATCH_KIND(env->kind, {
        auto kern = ilqr_solve_kernel<KIND>;
        if ((rc = prep(kern, smem)) != TFMPC_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(B), dim3(kWave), smem, st, *env, *cfg, a);
    });
    return launched();
}

int tfmpc_boxqp_f32(int B, int m, const float *H, const float *q, const float *low, const float *high, const float *x0,
                    float *x, float *free_mask, int32_t *status, void *stream)
{
    if (B < 0 || m <= 0) return TFMPC_ERR_ARG;
    if (B == 0) return TFMPC_OK;
    if (!H || !q || !low || !high || !x0 || !x) return TFMPC_ERR_ARG;
    if (m == 2 && !option_is(kOptIlqrKernel, "wave"))       // the lane kernels' own QP (closed form for two variables)
        return boxqp_lane2_launch(B, H, q, low, high, x0, x, free_mask, status, static_cast<hipStream_t>(stream));
    const size_t smem = (ilqr_smem_floats(1, m) + (size_t)m * odd_ld(m) + m) * sizeof(float);
    if (smem > kMaxLdsBytes) return TFMPC_ERR_UNSUPPORTED;
    int rc = prep(boxqp_kernel, smem);
    if (rc != TFMPC_OK) return rc;
    hipLaunchKernelGGL(boxqp_kernel, dim3(B), dim3(kWave), smem, static_cast<hipStream_t>(stream), m, H, q, low, high,
                       x0, x, free_mask, status);
    return launched();
}

}  // extern "C"
