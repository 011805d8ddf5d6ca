#!/usr/bin/env python3
"""Summarises the rocprofv3 passes written by tools/pmc_passes.sh into one JSON:
per-launch averages of every counter for the dominant kernel, HBM traffic with the
gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE doubled for wide coalesced
reads; both counters are in KiB), and the derived MFMA-busy fraction."""

import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
summary = {}

stats = glob.glob(os.path.join(out, "stats", "*", "*_kernel_stats.csv"))
kernel = None
if stats:
    # one file per profiled process: keep the one that spent the most GPU time (bench.py itself)
    tables = [list(csv.DictReader(open(f))) for f in stats]
    rows = max(tables, key=lambda t: sum(float(r["TotalDurationNs"]) for r in t))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    kernel = rows[0]["Name"]
    summary["kernel"] = kernel
    summary["calls"] = int(rows[0]["Calls"])
    summary["avg_duration_us"] = float(rows[0]["AverageNs"]) / 1e3

counters = defaultdict(list)
for f in glob.glob(os.path.join(out, "pass*", "*", "*_counter_collection.csv")):
    per_dispatch = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kernel and r["Kernel_Name"] != kernel:
            continue
        per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, name), v in per_dispatch.items():
        counters[name].append(v)
avg = {k: sum(v) / len(v) for k, v in counters.items()}
summary["counters_per_launch"] = avg

if "FETCH_SIZE" in avg or "WRITE_SIZE" in avg:
    fetch = avg.get("FETCH_SIZE", 0.0) * 1024.0
    write = avg.get("WRITE_SIZE", 0.0) * 1024.0
    summary["hbm_bytes_per_launch"] = {"fetch_raw": fetch, "fetch_x2_gfx950": 2 * fetch, "write": write,
                                       "total_corrected": 2 * fetch + write}
if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CU_CYCLES" in avg:
    # MFMA_BUSY counts cycles summed over SIMDs; BUSY_CU_CYCLES cycles summed over CUs (x4 SIMDs)
    summary["mfma_busy_frac_of_simd_time"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * avg["SQ_BUSY_CU_CYCLES"])
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import source_stamp
summary["csrc_sha16"] = source_stamp.stamp()      # the kernel sources this profile belongs to (bench.py checks it)
summary["batch_per_launch"] = 65536
summary["note"] = ("rocprofv3 passes of tools/pmc_passes.sh on bench.py (B=65536 per launch, --no-extra): the kernel-trace "
                   "stats pass runs bench.py's default step counts, the --pmc passes 4 steps; FETCH_SIZE/WRITE_SIZE in KiB, "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 wide coalesced reads)")
print(json.dumps(summary, indent=1))
