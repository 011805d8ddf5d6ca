"""Receding-horizon MPC episodes: the eager loop (Python + a handful of launches per control step)
against the same episode captured as ONE hipGraph (runners.Runner.capture).  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import agents, runners
from tfmpc.envs.navigation import Navigation
from tfmpc.solvers.ilqr import iLQR

T = 20
for B in (1, 64, 1024, 16384):
    rng = np.random.default_rng(1)
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    noise = [np.clip(rng.normal(0.0, 0.2, size=(B, 2, 1)), -0.4, 0.4).astype(np.float32) for _ in range(T)]
    for warm in (False, True):
        env = Navigation.load(problems.NAV_CONFIG); env.inject_noise(noise)
        agent = agents.MPC(iLQR(env), T, warm_start=warm, seed=3)
        runner = runners.Runner(env, agent)
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with runner(x0, T) as r: traj = r.run()
            torch.cuda.synchronize(); eager = time.perf_counter() - t0
        env2 = Navigation.load(problems.NAV_CONFIG)
        episode = runners.Runner(env2, agents.MPC(iLQR(env2), T, warm_start=warm, seed=3)).capture(x0, T, noise)
        x0d = torch.as_tensor(x0, device="cuda")
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            episode.x0.copy_(x0d); episode.graph.replay()
            torch.cuda.synchronize(); graphed = time.perf_counter() - t0
        print(f"navigation B={B:6d} T={T} warm_start={warm!s:5}: eager {eager / T * 1e3:6.3f} ms per control step, "
              f"one hipGraph {graphed / T * 1e3:6.3f} ms ({eager / graphed:.2f}x)", flush=True)
