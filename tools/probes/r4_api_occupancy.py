"""Times iLQR.solve on the LQ env (bench.py's ilqr_api workload generator) at T = 20 and T = 50 for the loaded build (TFMPC_LIB)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
B = 65536
msg = []
for T in (20, 50):
    w = workloads.ilqr_api_warm(B, horizon=T)
    s = workloads.solver_of(w)
    out = s.solve_device(w["x0"], T, u_init=w["u0"]); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); out = s.solve_device(w["x0"], T, u_init=w["u0"], workspace=out["workspace"]); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    its = float((out["iterations"].double() + 1).sum())
    msg.append(f"T={T}: {min(ts):.3f} ms, {its / B:.2f} iterations, {min(ts) * 1e6 / (its * T):.2f} ns per instance-iteration-step")
print(" | ".join(msg))
