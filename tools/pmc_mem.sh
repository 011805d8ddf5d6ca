#!/bin/bash
# Memory-side PMC passes (each in its own run, no other tracing beside --pmc) for the kernels of one python script:
#   tools/pmc_mem.sh <tag> <kernel-substring> <script.py> [args...]  -> gpurun_out/pmcmem_<tag>/summary.json
set -u
TAG=$1; MATCH=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcmem_$TAG
mkdir -p "$OUT"
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for C in "MemUnitBusy MemUnitStalled WriteUnitStalled" \
         "TCC_EA_WRREQ_STALL_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr" \
         "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
         "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pass$i" -- python3 $SCRIPT "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$MATCH" <<'PY' > "$OUT/summary.json"
import csv, glob, json, os, sys
from collections import defaultdict
out, match = sys.argv[1], sys.argv[2]
res = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "*", "*_counter_collection.csv")):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]: continue
        per[(r["Kernel_Name"][:100], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (kern, _, name), v in per.items(): res[kern][name].append(v)
print(json.dumps({k: {n: sum(v) / len(v) for n, v in d.items()} for k, d in res.items()}, indent=1))
PY
cat "$OUT/summary.json"; tail -3 "$OUT"/pass*.log | grep -i -B1 -A2 "error\|invalid\|not found" | head -20
