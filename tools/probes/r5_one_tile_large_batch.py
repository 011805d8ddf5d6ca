"""One-tile costate kernels at batch sizes that take the ONE-wave form (n = 16: one instance per column, 4 096 groups at B = 65 536): device time per solve,
median of 7 -- the check that the round-5 changes to the shared rollout / sweep code did not cost the large-batch forms anything (tools/probes/r5_rev_ab.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
T = 100
rng = np.random.default_rng(4)
for kind, n, B in (("hvac", 16, 65536), ("reservoir", 16, 65536), ("reservoir", 12, 32768), ("hvac", 10, 32768)):
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=5)
    out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{kind} n={n} B={B}: {sorted(ts)[3]:.3f} ms (min {min(ts):.3f})  cost {float(out['costs'].sum(dim=1).mean()):.6g}")
