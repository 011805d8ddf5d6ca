#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats plus PMC counter passes for
# bench.py's dominant kernel.  Counters are collected in their own passes (no
# sys/hip/hsa tracing beside --pmc), as MI355X_MICROARCH.md prescribes.
# usage: tools/pmc_passes.sh <tag>        -> gpurun_out/pmc_<tag>/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra"
# the stats pass runs bench.py's DEFAULT step counts, so its average kernel duration is the one
# bench.py itself reports (short runs clock lower: the first launches ramp)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra > "$OUT/stats.log" 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES" \
         "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pass$i" -- $BENCH > "$OUT/pass$i.log" 2>&1
done
python3 $ROOT/tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
