"""Summary of one tools/pmc_workload.sh directory: python tools/pmc_workload_summary.py <dir> <kernel-substring>  -> JSON on stdout."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out, match = sys.argv[1], sys.argv[2]
lines = [json.loads(l.split("PROFILE_WORKLOAD ", 1)[1]) for l in open(os.path.join(out, "stats.log")) if "PROFILE_WORKLOAD " in l]
# trace pass: per (kernel, grid) the dispatches in order
disp = defaultdict(list)
for f in glob.glob(os.path.join(out, "stats", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"]:
            disp[(r["Kernel_Name"][:120], int(r["Grid_Size_X"]))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
stats = {}
for (name, grid), ds in disp.items():
    ds.sort()
    dur = [(e - s) / 1e6 for s, e in ds]
    # the workload whose launch count this kernel's dispatch count fits (a workload makes warm + timed launches of its kernel; kernels that a
    # launch runs several times -- probes, second-chance launches -- are listed with what they are)
    timed = next((w["timed"] for w in lines if w["warm"] + w["timed"] == len(ds)), None)
    entry = {"grid": grid, "calls": len(ds), "avg_ms_all_dispatches": sum(dur) / len(dur), "min_ms": min(dur), "max_ms": max(dur)}
    if timed:
        last = ds[-timed:]
        entry["timed_avg_ms"] = sum(dur[-timed:]) / timed
        entry["timed_span_ms_per_launch"] = (last[-1][1] - last[0][0]) / 1e6 / timed       # incl. the gaps between the launches: what an event pair sees
    stats[f"{name} [grid {grid}]"] = entry
res = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "*", "*_counter_collection.csv")):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]:
            continue
        per[(f"{r['Kernel_Name'][:120]} [grid {int(float(r['Grid_Size']))}]", r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (kern, _, name), v in per.items():
        res[kern][name].append(v)
summary = {k: {n: sum(v) / len(v) for n, v in d.items()} for k, d in res.items()}
for k, d in summary.items():
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:        # KiB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
        fetch, write = d.get("FETCH_SIZE", 0.0) * 1024.0, d.get("WRITE_SIZE", 0.0) * 1024.0
        d["hbm_bytes_per_launch"] = {"fetch_raw": fetch, "fetch_x2_gfx950": 2 * fetch, "write": write, "total_corrected": 2 * fetch + write}
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "tools"))
try:
    import source_stamp
    src = source_stamp.stamp()
except Exception:
    src = None
print(json.dumps({"workloads": lines, "kernel_stats": stats, "counters_per_launch": summary, "csrc_sha16": src}, indent=1))
