// ilqr_lq_mfma.h -- launch interface of the matrix-core iLQR solve for the LQ env
// (ilqr_lq_mfma.hip).  Internal; the public contract is include/tfmpc_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tfmpc_hip.h"
#include "ilqr_trace.h"

namespace tfmpc {

// Internal status bit: "this instance needs the generic wave kernel" (non-PD Q_uu or a fully
// rejected line search, i.e. regularisation mu > 0).  Cleared by the second-chance launch and
// never visible to the caller.
constexpr int kIlqrRetryBit = 0x4000;

struct IlqrLqArgs {
    TfmpcEnv env;
    TfmpcIlqrConfig cfg;
    int B, T;
    const float *x0, *u_init;
    float *states, *actions, *costs;
    int32_t *iterations, *status;
    float *wsK, *wsk, *wsq;      // gains K[T][m][n], k[T][m] and Q_u[T][m] scratch (HBM)
    float *wsx, *wsu, *wsc;      // candidate trajectory x[T+1][n], u[T][m], costs[T+1] (ilqr_lq_mfma32.hip only)
    const int32_t *order;        // ilqr_lq_box_mfma.hip: block -> instance (heavy instances first), or null
    TraceArgs trace;             // optional decision trace (ilqr_trace.h): one row per backward pass + line search; rows == nullptr: none
    void *board;                 // ilqr_lq_box_mfma.hip: HBM for the helper teams' board (any alignment), or null: no helpers
    size_t board_bytes;
    int helper_teams, help_after;   // (set by the launcher) teams of helper blocks; passes an instance makes before it may claim one
    int speculate;                  // (set by the launcher) rejected passes in a row after which a team runs speculative sweeps; < 0: never
    const int32_t *gate;         // (set by the launcher) MODE 0 runs only if (*gate != 0) == (gate_value != 0); null: always
    int gate_value;
    float *wsMinv;               // ilqr_lq_mfma.hip / ilqr_lq_mfma32.hip: -Q_uu(t)^-1 [B][T][8][8] / [B][T][16][16] for the gain-reusing later passes, or null: full pass every time
};

// Control-limited twin (ilqr_lq_box_mfma.hip): bounded actions or any finite bound; the whole state machine
// (mu > 0, retries, rejections) in the kernel, so no second-chance launch.  Uses wsK, wsk, two ints per instance of wsq (block order) and `board`.
bool ilqr_lq_box_mfma_supported(const TfmpcEnv &env, int T);
int ilqr_lq_box_mfma_launch(const IlqrLqArgs &a, hipStream_t stream);

// Large-tile twin (ilqr_lq_mfma32.hip): unbounded LQ env beyond the 16 x 8 tile up to n = 32, m = 16; trajectories in HBM.
bool ilqr_lq_mfma32_supported(const TfmpcEnv &env, int T);
int ilqr_lq_mfma32_launch(const IlqrLqArgs &a, hipStream_t stream);

size_t ilqr_lq_mfma_lds_bytes(int T);
size_t ilqr_lq_mfma_reuse_workspace_bytes(int B, int n, int m, int T);     // the -Q_uu^-1 slab (0 for shapes the kernel does not take)
size_t ilqr_lq_mfma32_reuse_workspace_bytes(int B, int n, int m, int T);   // likewise for the large-tile twin ([T][16][16])
bool ilqr_lq_mfma_supported(const TfmpcEnv &env, int T);
int ilqr_lq_mfma_launch(const IlqrLqArgs &a, hipStream_t stream);

}  // namespace tfmpc
