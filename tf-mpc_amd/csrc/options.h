// options.h -- kernel-variant overrides for A/B timing and tests.  The environment variables
// TFMPC_LQR_KERNEL, TFMPC_LQR_MFMA, TFMPC_ILQR_KERNEL, TFMPC_COSTATE_WAVES, TFMPC_ILQR_RETRY, TFMPC_COSTATE_COUPLING, TFMPC_LQR_WAVES, TFMPC_BOX_HELPERS, TFMPC_BOX_HELP_AFTER, TFMPC_ILQR_LQ_REUSE, TFMPC_GROUP_STORED, TFMPC_BOX_SPECULATE are read ONCE per process (first launch);
// afterwards only tfmpc_set_option (include/tfmpc_hip.h) changes them.  Launchers compare by value:
//     if (option_is(kOptIlqrKernel, "wave")) ...
#pragma once

namespace tfmpc {

enum Option { kOptLqrKernel = 0, kOptLqrMfma = 1, kOptIlqrKernel = 2, kOptCostateWaves = 3, kOptIlqrRetry = 4, kOptCostateCoupling = 5, kOptLqrWaves = 6, kOptBoxHelpers = 7, kOptBoxHelpAfter = 8, kOptIlqrLqReuse = 9, kOptGroupStored = 10, kOptBoxSpeculate = 11, kOptCount = 12 };

// true when the option is set and equals `value`
bool option_is(Option which, const char *value);
// the option as an integer, `fallback` when unset or not a number
int option_int(Option which, int fallback);

}  // namespace tfmpc
