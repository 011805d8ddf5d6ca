"""bench.py's `control_limited.stable_open_loop_variant` workload (tests/workloads.py:control_limited_stable, 65 536 instances), two
launches, for profiling:   rocprofv3 --kernel-trace --stats -- python3 tools/box_stable_once.py     /   tools/pmc_kernel.sh box_stable ilqr_lq_box_mfma tools/box_stable_once.py
BOX_WORKLOAD=control_limited profiles the 0.25 F workload instead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
w = getattr(workloads, os.environ.get("BOX_WORKLOAD", "control_limited_stable"))(65536)
s = workloads.solver_of(w)
out = s.solve_device(w["x0"], w["T"], u_init=w["u0"]); torch.cuda.synchronize()
t0 = time.perf_counter()
out = s.solve_device(w["x0"], w["T"], u_init=w["u0"], workspace=out["workspace"]); torch.cuda.synchronize()
dt = time.perf_counter() - t0
it = (out["iterations"] + 1).float()
import hashlib
sha = hashlib.sha256(out["states"].cpu().numpy().tobytes() + out["actions"].cpu().numpy().tobytes() + out["iterations"].cpu().numpy().tobytes()).hexdigest()[:12]
print(f"{w['version']}: {dt * 1e3:.2f} ms per 65 536 solves, sha {sha}, mean iterations {float(it.mean()):.2f}, p50/p90/p99/p99.9/max "
      f"{[float(torch.quantile(it, q)) for q in (0.5, 0.9, 0.99, 0.999, 1.0)]}, status != 0: {int((out['status'] != 0).sum())}")
