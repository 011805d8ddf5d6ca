"""BASELINE configs[3] (Navigation iLQR, n = m = 2, T = 50, B = 16 384): one warm-up + 3 launches, for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.navigation import Navigation
from tfmpc.solvers.ilqr import iLQR
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
solver = iLQR(Navigation.load(problems.NAV_CONFIG))
x0 = np.random.default_rng(4).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
u0 = solver.random_actions(50, B, seed=4)
out = solver.solve_device(x0, 50, u_init=u0)
import time
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); out = solver.solve_device(x0, 50, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("iterations", float((out["iterations"].double() + 1).sum()), "| ms per batch:", " ".join(f"{t:.2f}" for t in ts))
