"""One-off: random shapes of every solver family with the caller's workspace pre-filled with NaN vs zeros (the fixed-shape
version is tests/test_workspace_independence_gpu.py).  python tools/probes/fuzz_workspace.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.lq import LQEnv
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
from tfmpc.solvers.lqr import LQR
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(5)
KEYS = ("states", "actions", "costs", "iterations", "status")
def same(a, b): return torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))
bad = 0
for case in range(cases):
    fam = ("costate", "lq", "lqr", "nav")[case % 4]
    B, T = int(rng.choice([1, 2, 3, 5, 17, 33, 63, 65, 130, 700])), int(rng.integers(1, 40))
    if fam == "costate":
        n = int(rng.integers(1, 33)); kind = ("hvac", "reservoir")[int(rng.integers(2))]
        env = (HVAC.load(dict(problems.hvac_config(n, seed=case))) if kind == "hvac" else Reservoir.load(dict(problems.reservoir_config(n, seed=case))))
        x0 = rng.uniform(20.0, 60.0, size=(B, n, 1)).astype(np.float32)
        s = iLQR(env, max_iterations=4, storage_bf16=bool(rng.integers(2))); u0 = s.random_actions(T, B, seed=case)
        solve = lambda ws: s.solve_device(x0, T, u_init=u0, workspace=ws); tag = f"{kind} n={n}"
    elif fam == "lq":
        n, m = int(rng.integers(3, 33)), int(rng.integers(1, 17)); bound = None if rng.integers(2) else 0.7
        F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=case)
        low, high = (None, None) if bound is None else (-bound, bound)
        s = iLQR(LQEnv(F * 0.2 * np.sqrt(16.0 / n), f, C, c, low=low, high=high), max_iterations=4)
        u0 = np.zeros((B, T, m, 1), dtype=np.float32); x0 = x0.astype(np.float32)[..., None]
        solve = lambda ws: s.solve_device(x0, T, u_init=u0, workspace=ws); tag = f"lq n={n} m={m} bound={bound}"
    elif fam == "lqr":
        n, m = int(rng.integers(1, 41)), int(rng.integers(1, 25))
        F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=case)
        lqr = LQR(F * 0.5 * 2 / np.sqrt(n), f, C, c); x0d = lqr._prep_x0(x0)
        solve = lambda ws: lqr.solve_device(x0d, T, workspace=ws); tag = f"lqr n={n} m={m}"
    else:
        env = Navigation.load(problems.NAV_CONFIG); x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
        s = iLQR(env, max_iterations=5); u0 = s.random_actions(T, B, seed=case)
        solve = lambda ws: s.solve_device(x0, T, u_init=u0, workspace=ws); tag = "navigation"
    first = solve(None); torch.cuda.synchronize(); ws = first["workspace"]
    outs = []
    for fill in (0.0, float("nan")):
        ws.fill_(fill); o = solve(ws); torch.cuda.synchronize()
        outs.append({k: o[k].clone() for k in KEYS if k in o})
    ok = all(same(outs[0][k], outs[1][k]) for k in outs[0])
    bad += not ok
    print(f"case {case:3d} {tag} T={T} B={B}: {'ok' if ok else 'DEPENDS ON WORKSPACE CONTENTS'}", flush=True)
print("failures:", bad); sys.exit(1 if bad else 0)
