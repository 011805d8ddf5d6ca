"""A/B of the control-limited kernel: the loaded build (TFMPC_LIB or the product) on bench.py's two control-limited workloads; prints times and a
checksum of every output so that two runs (two builds) can be compared bit for bit:   python tools/probes/r5_box_pair_ab.py"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
for name in ("control_limited_stable", "control_limited"):
    w = getattr(workloads, name)(65536)
    s = workloads.solver_of(w)
    out = s.solve_device(w["x0"], w["T"], u_init=w["u0"], trace_rows=170); torch.cuda.synchronize()
    h = hashlib.sha256()
    for k in ("states", "actions", "costs", "iterations", "status", "trace_len"):
        h.update(out[k].cpu().numpy().tobytes())
    h.update(torch.nan_to_num(out["trace"]).cpu().numpy().tobytes())
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); o2 = s.solve_device(w["x0"], w["T"], u_init=w["u0"], workspace=out["workspace"]); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: {min(ts):.2f} ms (runs {' '.join(f'{t:.1f}' for t in ts)}), outputs + trace sha {h.hexdigest()[:16]}", flush=True)
