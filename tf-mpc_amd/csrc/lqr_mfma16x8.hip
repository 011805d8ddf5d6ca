// lqr_mfma16x8.hip -- placeholder until the MFMA variant lands: reports
// "unsupported" so the dispatcher uses the generic wave kernel.
#include "lqr_kernels.h"

namespace tfmpc {
bool lqr_mfma_supported(int, int) { return false; }
int lqr_mfma_launch(const LqrArgs &, bool, bool, hipStream_t) { return TFMPC_ERR_UNSUPPORTED; }
}  // namespace tfmpc
