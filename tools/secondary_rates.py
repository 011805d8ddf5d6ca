"""Throughput of the BASELINE configs that are NOT bench lines (cfg2, cfg4, cfg5) and of the
headline shape through the iLQR API.  Run on the GPU box: python tools/secondary_rates.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.envs.lq import LQEnv
from tfmpc.envs import make_lqr_linear_navigation
from tfmpc.solvers.ilqr import iLQR

def run(name, solver, x0, T, u0, reps=3):
    out = solver.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    its = (out["iterations"].double() + 1).sum().item()
    print(f"{name}: B={x0.shape[0]} T={T} {dt*1e3:.2f} ms/solve-batch, mean iters {its/x0.shape[0]:.1f}, "
          f"{its/dt:.3e} iLQR iterations/s, flagged {(out['status']!=0).sum().item()}")

rng = np.random.default_rng(4)
env = Navigation.load(problems.NAV_CONFIG); s = iLQR(env)
B, T = 16384, 50
x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32); u0 = s.random_actions(T, B, seed=4)
run("cfg4 navigation", s, x0, T, u0)
# cfg2 via LQR path
F, f, C, c, x0n, goal = problems.make_navlin_batch(4096, 5.0)
lqr = make_lqr_linear_navigation(goal[..., None], 5.0)
o = lqr.solve_device(x0n[..., None], 50); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): o = lqr.solve_device(x0n[..., None], 50, workspace=o["workspace"])
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print(f"cfg2 navlin LQR: B=4096 T=50 {dt*1e3:.3f} ms, {4096/dt:.3e} solves/s")
# headline shape through the iLQR path (LQ env, stable-ish F)
Fb, fb, Cb, cb, xb = problems.make_lqr_batch_fast(8192, 16, 8, seed=1)
Fb *= 0.25
s = iLQR(LQEnv(Fb, fb, Cb, cb)); u0 = torch.zeros(8192, 50, 8, 1, device="cuda")
run("iLQR on LQ env n=16 m=8", s, xb[..., None].astype(np.float32), 50, u0)
for kind in ("hvac", "reservoir"):
    n, T, B = 32, 100, 32768
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=5)
    run(f"cfg5 {kind} n=32", s, x0, T, u0, reps=1)
# headline shape through the iLQR API on the matrix cores, full batch
Fb, fb, Cb, cb, xb = problems.make_lqr_batch_fast(65536, 16, 8, seed=1)
Fb *= 0.25
s = iLQR(LQEnv(Fb, fb, Cb, cb)); u0 = torch.zeros(65536, 50, 8, 1, device="cuda")
run("iLQR API, LQ env n=16 m=8 (matrix cores)", s, xb[..., None].astype(np.float32), 50, u0, reps=5)
