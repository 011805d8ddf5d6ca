"""cfg5 (HVAC / Reservoir n = m = 32, T = 100): where an iLQR iteration's time goes.  Solves with the
full 11-point line search and with a 1-point one (every iteration = one backward + one rollout), at
several iteration caps.  Run on the GPU box: python tools/cfg5_breakdown.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

n, T, B = 32, 100, 8192
rng = np.random.default_rng(4)
for kind in ("hvac", "reservoir"):
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
    for one_alpha in (False, True):
        for iters in (1, 4, 12):
            s = iLQR(env, max_iterations=iters)
            if one_alpha:
                s._alphas = lambda: np.array([1.0])
            u0 = s.random_actions(T, B, seed=5)
            out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
            t = time.perf_counter()
            out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
            dt = time.perf_counter() - t
            print(f"{kind} alphas={'1' if one_alpha else '11'} max_iterations={iters}: {dt*1e3:.1f} ms, mean iterations "
                  f"{(out['iterations'].double()+1).mean().item():.1f}", flush=True)
