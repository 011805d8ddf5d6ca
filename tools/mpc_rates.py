"""Receding-horizon MPC episodes batched over episodes: wall time per control step, split
into the iLQR re-solve and the rest (env step, noise, Python).  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import agents, runners
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

for name, env, x0f, T in (("navigation", Navigation.load(problems.NAV_CONFIG), lambda B: np.random.default_rng(1).uniform(0, 10, size=(B, 2, 1)).astype(np.float32), 20),
                          ("res4", Reservoir.load(dict(problems.RES4_CONFIG)), lambda B: np.tile(np.array(problems.RES4_X0, dtype=np.float32)[None], (B, 1, 1)), 20)):
    for B in (1, 1024, 16384):
        for warm in (False, True):
            solver = iLQR(env)
            agent = agents.MPC(solver, T, warm_start=warm, seed=3)
            runner = runners.Runner(env, agent)
            env.seed(3)
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                with runner(x0f(B), T) as r:
                    traj = r.run()
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            its = np.mean([np.mean(i) for i in agent.iterations])
            print(f"{name:10s} B={B:5d} T={T} warm_start={warm!s:5}: {dt*1e3:8.1f} ms per episode batch, {dt/T*1e3:6.2f} ms per control step, "
                  f"mean iLQR iterations per re-solve {its:.1f}, mean total cost {float(np.mean(traj.total_cost)):.2f}")
