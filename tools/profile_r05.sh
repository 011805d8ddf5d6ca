#!/bin/bash
# Round-5 measurements on the GPU box (copy what is to be judged into profiles/):  bash tools/profile_r05.sh <part>
#   part a: headline PMC passes (tools/pmc_passes.sh) + iLQR API + control-limited (0.25 F and the stable variant)
#   part b: cfg5 (HVAC + the Reservoir chain instantiation), small envs, large tiles, cfg4
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export CFG5_ONCE_SINGLE=1
case "${1:-a}" in
a)
  bash tools/pmc_passes.sh r05 > gpurun_out/r05_headline_pmc.log 2>&1; echo headline done
  bash tools/pmc_kernel.sh r05_ilqr_api ilqr_lq_mfma_kernel tools/ilqr_api_once.py > /dev/null 2>&1; echo api done
  bash tools/pmc_kernel.sh r05_box_stable ilqr_lq_box_mfma tools/box_stable_once.py > /dev/null 2>&1; echo box stable done
  bash tools/pmc_kernel.sh r05_box ilqr_lq_box_mfma tools/ilqr_api_once.py box > /dev/null 2>&1; echo box done
  ;;
b)
  bash tools/pmc_kernel.sh r05_cfg5 ilqr_adjoint_mfma tools/cfg5_once.py > /dev/null 2>&1; echo cfg5 done
  bash tools/pmc_kernel.sh r05_small_env ilqr_adjoint_mfma tools/small_env_once.py > /dev/null 2>&1; echo small done
  bash tools/pmc_kernel.sh r05_large_tile mfma32 tools/large_tile_once.py > /dev/null 2>&1; echo large done
  bash tools/pmc_kernel.sh r05_cfg4 ilqr_group_solve tools/cfg4_once.py > /dev/null 2>&1; echo cfg4 done
  ;;
esac
ls gpurun_out | grep pmc_r05
