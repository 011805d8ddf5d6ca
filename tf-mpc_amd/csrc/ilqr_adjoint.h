// ilqr_adjoint.h -- launch interface of the register-resident HVAC / Reservoir solve
// (ilqr_adjoint.hip).  Internal; the public contract is include/tfmpc_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tfmpc_hip.h"
#include "ilqr_trace.h"

namespace tfmpc {

struct AdjointSolveArgs {
    int B, T;
    const float *x0, *u_init;
    float *states, *actions, *costs;
    int32_t *iterations, *status;
    float *wsk, *wsx, *wsu, *wsc;        // gains k[T][m], candidate x[T+1][n], u[T][m], costs[T+1]
    void *wave_ws;                       // 16-per-wave kernel: its wave-major trajectory buffers (ilqr_adjoint_mfma_workspace_bytes)
    TraceArgs trace;                     // 16-per-wave kernel only: the optional decision trace
    int dense_coupling;                  // 16-per-wave kernel: 1 = multiply by the coupling matrix even when it is a shift (A/B runs, tests)
};

bool ilqr_adjoint_supported(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg);
int ilqr_adjoint_launch(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, const AdjointSolveArgs &a, hipStream_t stream);

// sixteen instances per wave, coupling-matrix products on the matrix cores (ilqr_adjoint_mfma.hip): envs whose
// parameters are shared by the whole batch
bool ilqr_adjoint_mfma_supported(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg);
int ilqr_adjoint_mfma_launch(const TfmpcEnv &env, const TfmpcIlqrConfig &cfg, const AdjointSolveArgs &a, hipStream_t stream);
// bytes of AdjointSolveArgs::wave_ws for a batch of B instances with n states (0 for shapes the kernel does not serve)
size_t ilqr_adjoint_mfma_workspace_bytes(int B, int n, int m, int T);

}  // namespace tfmpc
