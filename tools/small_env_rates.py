"""iLQR on the reference's own small env configs (hvac6.config.json, res4.config.json: n = 6 / 4) at large batch.
Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

rng = np.random.default_rng(0)
for name, env, x0 in (("hvac6", HVAC.load(dict(problems.HVAC6_CONFIG)), np.array(problems.HVAC6_X0, dtype=np.float32)),
                      ("res4", Reservoir.load(dict(problems.RES4_CONFIG)), np.array(problems.RES4_X0, dtype=np.float32))):
    for B, T in ((16384, 100), (65536, 40)):
        x = (x0[None] * rng.uniform(0.9, 1.1, size=(B, 1, 1))).astype(np.float32)
        s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=1)
        out = s.solve_device(x, T, u_init=u0); torch.cuda.synchronize()
        ts = []
        for _ in range(9):                      # (one launch each; device time between two events; the median is quoted)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = s.solve_device(x, T, u_init=u0, workspace=out["workspace"]); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        dt = sorted(ts)[len(ts) // 2]
        its = (out["iterations"].double() + 1).sum().item()
        print(f"{name}: B={B} T={T}: {dt*1e3:.3f} ms (min {min(ts)*1e3:.3f}), mean iterations {its/B:.1f}, {its/dt:.3e} iterations/s, flagged {(out['status']!=0).sum().item()}")
