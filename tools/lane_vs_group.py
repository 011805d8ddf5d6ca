"""cfg4 (Navigation n=m=2, T=50): one-lane-per-instance vs 16-lanes-per-instance (parallel line
search) fused iLQR kernels over batch size.  Run on the GPU box: python tools/lane_vs_group.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.navigation import Navigation
from tfmpc import _hip
from tfmpc.solvers.ilqr import iLQR

env = Navigation.load(problems.NAV_CONFIG); s = iLQR(env); T = 50
for B in (64, 1024, 4096, 16384, 65536, 262144):
    rng = np.random.default_rng(4)
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32); u0 = s.random_actions(T, B, seed=4)
    row = []
    for kern in ("lane", "lane1"):
        _hip.set_option("TFMPC_ILQR_KERNEL", kern)
        out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
        torch.cuda.synchronize(); row.append((time.perf_counter() - t) / 3 * 1e3)
    it = out["iterations"].cpu().numpy() + 1
    print(f"B={B}: group {row[0]:.2f} ms, per-lane {row[1]:.2f} ms; iterations mean {it.mean():.1f} "
          f"p50 {np.median(it):.0f} p99 {np.quantile(it, .99):.0f} max {it.max()}", flush=True)
