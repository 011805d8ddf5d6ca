"""The reference's own env configs (hvac6.config.json n = 6, res4.config.json n = 4) at B = 16 384, T = 100, 12 iterations: one
warm-up + one launch each, for profiling (rocprofv3 ... -- python3 tools/small_env_once.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
B, T = 16384, 100
for name, env, x0r in (("hvac6", HVAC.load(dict(problems.HVAC6_CONFIG)), problems.HVAC6_X0),
                       ("res4", Reservoir.load(dict(problems.RES4_CONFIG)), problems.RES4_X0)):
    x0 = np.tile(np.array(x0r, dtype=np.float32)[None], (B, 1, 1))
    s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=1)
    out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    t = time.perf_counter(); out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) * 1e3:.2f} ms, {float((out['iterations'].double() + 1).sum()):.0f} iterations")
