"""What a drop-in user sees first: ONE problem instance (no batch axis) through the reference's API, wall time of
solver.solve(...) including Python, launch and the copy back.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs import make_lqr, make_lqr_linear_navigation
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR


def wall(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, out

np.random.seed(0)
lqr = make_lqr(3, 2)
ms, _ = wall(lambda: lqr.solve(np.array([[-1.0], [0.5], [3.6]], dtype=np.float32), 10))
print(f"tfmpc lqr (n=3, m=2, T=10, README example): {ms:.3f} ms per solve")
nav = make_lqr_linear_navigation(np.array([[8.0], [-9.0]], dtype=np.float32), 5.0)
ms, _ = wall(lambda: nav.solve(np.zeros((2, 1), dtype=np.float32), 10))
print(f"tfmpc navlin (T=10, README example): {ms:.3f} ms per solve")
for name, env, x0, T in (("navigation (nav.config.json)", Navigation.load(problems.NAV_CONFIG), np.array([[1.0], [1.5]], dtype=np.float32), 50),
                         ("hvac6", HVAC.load(dict(problems.HVAC6_CONFIG)), np.array(problems.HVAC6_X0, dtype=np.float32), 40),
                         ("res4", Reservoir.load(dict(problems.RES4_CONFIG)), np.array(problems.RES4_X0, dtype=np.float32), 40)):
    s = iLQR(env)
    ms, (traj, its) = wall(lambda: s.solve(x0, T, seed=1), reps=5)
    print(f"tfmpc ilqr {name}, T={T}: {ms:.2f} ms per solve, {its + 1} iterations, total cost {traj.total_cost:.3f}")
