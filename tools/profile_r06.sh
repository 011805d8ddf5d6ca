#!/bin/bash
# Round-6 measurements on the GPU box (copy what is to be judged into profiles/ with tools/collect_profiles.sh r06):  bash tools/profile_r06.sh <part>
# Every kernel-trace pass launches its workload >= 40 times untimed + 10 timed (tools/profile_workload.py: the SAME problems bench.py times,
# tests/workloads.py) and the summaries hold the profile's duration beside the tool's own; the long launches state their counts.
#   part a: headline PMC passes on bench.py itself (tools/pmc_passes.sh) + iLQR API (both forms)
#   part b: cfg5 x 2, hvac6, res4       part c: cfg4 (single batch, one launch of 131 072), cfg2, large tiles      part d: control-limited (stable: PMC; 0.25 F: durations only)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
W=tools/pmc_workload.sh
case "${1:-a}" in
a)
  bash tools/pmc_passes.sh r06 > gpurun_out/r06_headline_pmc.log 2>&1; echo headline done
  bash $W r06_ilqr_api ilqr_lq_mfma_kernel ilqr_api > /dev/null 2>&1; echo api done
  bash $W r06_ilqr_api_full ilqr_lq_mfma_kernel ilqr_api_full stats-only > /dev/null 2>&1; echo api full done
  ;;
b)
  bash $W r06_cfg5_hvac ilqr_adjoint_mfma cfg5_hvac > /dev/null 2>&1; echo cfg5 hvac done
  bash $W r06_cfg5_reservoir ilqr_adjoint_mfma cfg5_reservoir > /dev/null 2>&1; echo cfg5 reservoir done
  bash $W r06_hvac6 ilqr_adjoint_mfma hvac6 > /dev/null 2>&1; echo hvac6 done
  bash $W r06_res4 ilqr_adjoint_mfma res4 > /dev/null 2>&1; echo res4 done
  ;;
c)
  bash $W r06_cfg4 ilqr_group_solve cfg4 > /dev/null 2>&1; echo cfg4 done
  bash $W r06_cfg4_one_launch ilqr_group_solve cfg4_one_launch > /dev/null 2>&1; echo cfg4 one launch done
  bash $W r06_cfg2 lqr_lane cfg2 > /dev/null 2>&1; echo cfg2 done
  bash $W r06_lqr32 lqr_mfma32x16 lqr32 > /dev/null 2>&1; echo lqr32 done
  bash $W r06_literal_dims ilqr_lq_mfma32 literal_dims > /dev/null 2>&1; echo literal dims done
  ;;
d)
  bash $W r06_box_stable ilqr_lq_box_mfma box_stable > /dev/null 2>&1; echo box stable done
  bash $W r06_box ilqr_lq_box_mfma box stats-only > /dev/null 2>&1; echo box done
  ;;
esac
ls gpurun_out | grep pmc_r06
