"""One-off: captured MPC episodes (runners.Runner.capture) against the eager loop on random envs / batch sizes / horizons,
bit for bit (the fixed cases are tests/test_mpc_graph_gpu.py).  python tools/probes/fuzz_graph.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import agents, runners
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(11)
bad = 0
def make(kind, n):
    if kind == "navigation": return Navigation.load(problems.NAV_CONFIG)
    if kind == "reservoir": return Reservoir.load(dict(problems.reservoir_config(n, seed=3)))
    return HVAC.load(dict(problems.hvac_config(n, seed=3)))
for case in range(cases):
    kind = ("navigation", "reservoir")[case % 2]          # (HVAC has no GymEnv stepping in the reference either)
    n = 2 if kind == "navigation" else int(rng.integers(2, 13))
    B, T, warm = int(rng.choice([1, 3, 20, 100])), int(rng.integers(2, 9)), bool(rng.integers(2))
    def inputs():
        if kind == "navigation":
            return rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32), [np.clip(rng.normal(0, 0.2, size=(B, 2, 1)), -0.4, 0.4).astype(np.float32) for _ in range(T)]
        if kind == "reservoir":
            return rng.uniform(30, 60, size=(B, n, 1)).astype(np.float32), [rng.gamma(2.0, 1.0, size=(B, n, 1)).astype(np.float32) for _ in range(T)]
        return rng.uniform(10, 25, size=(B, n, 1)).astype(np.float32), [np.zeros((B, n, 1), dtype=np.float32) for _ in range(T)]
    a, b = inputs(), inputs()
    env = make(kind, n)
    episode = runners.Runner(env, agents.MPC(iLQR(env, max_iterations=10), T, warm_start=warm, seed=5)).capture(a[0], T, a[1])
    ok = True
    for x0, noise in (a, b):
        traj, its = episode(x0, noise)
        env2 = make(kind, n); env2.inject_noise(noise)
        agent2 = agents.MPC(iLQR(env2, max_iterations=10), T, warm_start=warm, seed=5)
        with runners.Runner(env2, agent2)(x0, T) as r: ref = r.run()
        ref_its = np.stack([np.asarray(i).reshape(-1) for i in agent2.iterations])
        ok = ok and np.array_equal(traj.states, ref.states) and np.array_equal(traj.actions, ref.actions) and np.array_equal(traj.costs, ref.costs) and np.array_equal(its, ref_its)
    bad += not ok
    print(f"case {case:2d} {kind:10s} n={n:2d} B={B:3d} T={T} warm={warm}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("failures:", bad); sys.exit(1 if bad else 0)
