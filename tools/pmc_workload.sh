#!/bin/bash
# rocprofv3 kernel trace + PMC counter passes (each in its own run, no other tracing beside --pmc) for ONE group of bench.py's workloads
# (tools/profile_workload.py: the same problems bench.py times, 40 untimed + 10 timed launches in the trace pass):
#   tools/pmc_workload.sh <tag> <kernel-substring> <group> [stats-only]       -> gpurun_out/pmc_<tag>/summary.json
# summary.json: per kernel (name + grid size) the average duration of its LAST `timed` dispatches in the trace pass beside the ms per launch
# the tool itself measured in the same process (one event pair around the timed launches), the all-dispatch average rocprofv3 --stats
# would print, and the per-launch counters.
set -u
TAG=$1; MATCH=$2; GROUP=$3; ONLY=${4:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
SCRIPT=$ROOT/tools/profile_workload.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $SCRIPT $GROUP > "$OUT/stats.log" 2>&1
if [ "$ONLY" != "stats-only" ]; then
  export PROFILE_WARM=2 PROFILE_TIMED=2
  i=0
  for C in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pass$i" -- python3 $SCRIPT $GROUP > "$OUT/pass$i.log" 2>&1
  done
  unset PROFILE_WARM PROFILE_TIMED
fi
GRAFT_REPO_ROOT=$ROOT python3 $ROOT/tools/pmc_workload_summary.py "$OUT" "$MATCH" > "$OUT/summary.json"
cat "$OUT/summary.json" | head -60
