#!/bin/bash
# PMC passes on the cfg5 costate kernels: tools/pmc_cfg5.sh <tag> -> gpurun_out/pmc_cfg5_<tag>/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_cfg5_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/cfg5_once.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
i=0
for C in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pass$i" -- $CMD > "$OUT/pass$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
res = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "*", "*_counter_collection.csv")):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "ilqr_adjoint" not in r["Kernel_Name"]: continue
        kind = "hvac" if "ILi3E" in r["Kernel_Name"] or "(3)" in r["Kernel_Name"] or "<3>" in r["Kernel_Name"] or "<3," in r["Kernel_Name"] else "reservoir"
        per[(kind, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (kind, _, name), v in per.items(): res[kind][name].append(v)
summary = {k: {n: sum(v) / len(v) for n, v in d.items()} for k, d in res.items()}
stats = {}
for f in glob.glob(os.path.join(out, "stats", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "ilqr_adjoint" in r["Name"]: stats[r["Name"][:90]] = float(r["AverageNs"]) / 1e6
print(json.dumps({"avg_ms": stats, "counters_per_launch": summary}, indent=1))
PY
