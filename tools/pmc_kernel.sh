#!/bin/bash
# rocprofv3 kernel-trace stats + PMC counter passes (each in its own run, no other tracing beside --pmc) for the
# kernels of ONE python script whose names contain a substring; per-launch averages as JSON.
#   tools/pmc_kernel.sh <tag> <kernel-substring> <script.py> [script args...]   -> gpurun_out/pmc_<tag>/summary.json
set -u
TAG=$1; MATCH=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $SCRIPT "$@" > "$OUT/stats.log" 2>&1
i=0
for C in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pass$i" -- python3 $SCRIPT "$@" > "$OUT/pass$i.log" 2>&1
done
GRAFT_REPO_ROOT=$ROOT python3 - "$OUT" "$MATCH" <<'PY' > "$OUT/summary.json"
import csv, glob, json, os, sys
from collections import defaultdict
out, match = sys.argv[1], sys.argv[2]
res = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "*", "*_counter_collection.csv")):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]: continue
        per[(r["Kernel_Name"][:120], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (kern, _, name), v in per.items(): res[kern][name].append(v)
summary = {k: {n: sum(v) / len(v) for n, v in d.items()} for k, d in res.items()}
for k, d in summary.items():
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:        # KiB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
        fetch, write = d.get("FETCH_SIZE", 0.0) * 1024.0, d.get("WRITE_SIZE", 0.0) * 1024.0
        d["hbm_bytes_per_launch"] = {"fetch_raw": fetch, "fetch_x2_gfx950": 2 * fetch, "write": write, "total_corrected": 2 * fetch + write}
stats = {}
for f in glob.glob(os.path.join(out, "stats", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if match in r["Name"]: stats[r["Name"][:120]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "tools"))
try:
    import source_stamp
    src = source_stamp.stamp()
except Exception:
    src = None
print(json.dumps({"kernel_stats": stats, "counters_per_launch": summary, "csrc_sha16": src}, indent=1))
PY
cat "$OUT/summary.json"
