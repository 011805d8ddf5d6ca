"""cfg4 (Navigation iLQR, n = m = 2, T = 50) in ONE launch: 16 384 instances (BASELINE configs[3]) and 8 x 16 384 = 131 072 instances
(what round 3 needed eight host streams for) through the persistent group kernel with its instance queue.  JSON to stdout.  GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.navigation import Navigation
from tfmpc.solvers.ilqr import iLQR
T = 50
solver = iLQR(Navigation.load(problems.NAV_CONFIG))
res = {}
for B in (16384, 131072):
    x0 = torch.as_tensor(np.concatenate([np.random.default_rng(4 if B == 16384 else 100 + i).uniform(0, 10, size=(16384, 2, 1)) for i in range(B // 16384)]).astype(np.float32), device="cuda")
    u0 = torch.cat([solver.random_actions(T, 16384, seed=4 if B == 16384 else 100 + i) for i in range(B // 16384)])
    out = solver.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    its = float((out["iterations"].double() + 1).sum())
    res[f"B={B}"] = {"ms_min": min(ts) * 1e3, "ms_median": float(np.median(ts)) * 1e3, "iterations_per_s_at_median": its / float(np.median(ts)),
                     "mean_iterations": its / B, "max_iterations": int(out["iterations"].max()) + 1, "flagged": int((out["status"] != 0).sum())}
print(json.dumps(res, indent=1))
