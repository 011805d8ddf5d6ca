#!/bin/bash
# Round-3 measurements on the GPU box (copy what is to be judged into profiles/):  bash tools/profile_r03.sh <part>
#   part a: headline PMC passes (tools/pmc_passes.sh) + cfg5 + cfg4          part b: iLQR API, small envs, N4 output bytes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export CFG5_ONCE_SINGLE=1
case "${1:-a}" in
a)
  bash tools/pmc_passes.sh r03 > gpurun_out/r03_headline_pmc.log 2>&1; echo headline done
  bash tools/pmc_kernel.sh r03_cfg5 ilqr_adjoint_mfma tools/cfg5_once.py > /dev/null 2>&1; echo cfg5 done
  bash tools/pmc_kernel.sh r03_cfg4 ilqr_group_solve tools/cfg4_once.py > /dev/null 2>&1; echo cfg4 done
  ;;
b)
  bash tools/pmc_kernel.sh r03_ilqr_api ilqr_lq_mfma_kernel tools/ilqr_api_once.py > /dev/null 2>&1; echo api done
  bash tools/pmc_kernel.sh r03_small_env ilqr_adjoint_mfma tools/small_env_once.py > /dev/null 2>&1; echo small done
  bash tools/pmc_kernel.sh r03_lqr_out_f32 lqr_mfma16x8 tools/lqr_outputs_once.py > /dev/null 2>&1; echo out32 done
  bash tools/pmc_kernel.sh r03_lqr_out_bf16 lqr_mfma16x8 tools/lqr_outputs_once.py bf16 > /dev/null 2>&1; echo out16 done
  ;;
c)
  bash tools/pmc_kernel.sh r03_cfg5 ilqr_adjoint_mfma tools/cfg5_once.py > /dev/null 2>&1; echo cfg5 done
  bash tools/pmc_kernel.sh r03_small_env ilqr_adjoint_mfma tools/small_env_once.py > /dev/null 2>&1; echo small done
  unset CFG5_ONCE_SINGLE
  python tools/cfg4_sustained.py > gpurun_out/r03_cfg4_sustained.json 2> /dev/null; echo cfg4 sustained done
  bash tools/probes/r3_cfg5_libs.sh lib_r02.so product > gpurun_out/r03_cfg5_ab.txt 2>&1; echo ab done
  TFMPC_LIB=$ROOT/tools/probes/ab/lib_probe.so python tools/probes/cfg5_phases.py 32768 16384 2>&1 | grep -v amdgpu > gpurun_out/r03_cfg5_phase_split.txt; echo phases done
  tools/probes/matvec_probe > gpurun_out/r03_matvec_probe.txt 2>&1
  python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; echo bench done
  ;;
esac
ls gpurun_out | grep pmc_r03
