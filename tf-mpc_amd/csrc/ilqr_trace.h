// ilqr_trace.h -- the optional decision trace of the fused iLQR solve kernels (tfmpc_ilqr_solve_trace_f32,
// include/tfmpc_hip.h: one row per backward pass + line search of an instance, ilqr.py:238-279).  Internal.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tfmpc_hip.h"

namespace tfmpc {

struct TraceArgs {
    float *rows;          // [B][max_rows][TFMPC_TRACE_COLS], or nullptr: no trace
    int32_t *len;         // [B]: passes made
    int max_rows;
    // tfmpc_ilqr_solve_trace_qp_f32 only (else nullptr): per pass and time step, what the box-QP of the controller ended on
    // (ilqr.py:364-385, optimization.py:35-72) -- written by the control-limited matrix-core kernel
    uint8_t *clamp;       // [B][max_rows][T]: bit a set = action a CLAMPED in the last factorised free set (row a of K_t is zero)
    uint8_t *qp_it;       // [B][max_rows][T]: iterations of the projected-Newton loop (optimization.py:24), 0 = no QP at this step
};

// called by ONE lane of the instance, once per pass, with row = the number of passes before this one
__device__ __forceinline__ void trace_write(const TraceArgs &t, size_t b, int row, int iteration, float mu, float delta,
                                            float J_hat, float g_norm, int alpha_index, float alpha, float J, int accepted,
                                            float residual, int level = 0)
{
    if (!t.rows) return;
    if (row < t.max_rows) {
        float *r = t.rows + (b * (size_t)t.max_rows + (size_t)row) * TFMPC_TRACE_COLS;
        r[TFMPC_TR_ITERATION] = (float)iteration;
        r[TFMPC_TR_MU] = mu;
        r[TFMPC_TR_DELTA] = delta;
        r[TFMPC_TR_J_HAT] = J_hat;
        r[TFMPC_TR_G_NORM] = g_norm;
        r[TFMPC_TR_ALPHA_INDEX] = (float)alpha_index;
        r[TFMPC_TR_ALPHA] = alpha;
        r[TFMPC_TR_J] = J;
        r[TFMPC_TR_ACCEPTED] = (float)accepted;
        r[TFMPC_TR_RESIDUAL] = residual;
        r[TFMPC_TR_LEVEL] = (float)level;
    }
    t.len[b] = row + 1;
}

}  // namespace tfmpc
