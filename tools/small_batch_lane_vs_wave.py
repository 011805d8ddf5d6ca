"""Tiny 2-D envs at SMALL batch: the 16-lanes-per-instance group kernel (TFMPC_ILQR_KERNEL=lane) vs the wave kernel
(=wave), wall time of solve_device.  Where should the dispatcher's batch threshold sit?  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.lqr.navigation import NavigationLQR
from tfmpc import _hip
from tfmpc.solvers.ilqr import iLQR

for name, env in (("navigation", Navigation.load(problems.NAV_CONFIG)), ("navigation_lqr bounded", NavigationLQR([[5.5], [-9.0]], 5.0, -1.0, 1.0))):
    for B in (1, 2, 4, 8, 16, 31):
        rng = np.random.default_rng(4)
        x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
        row = []
        for force in ("lane", "wave"):
            _hip.set_option("TFMPC_ILQR_KERNEL", force)
            s = iLQR(env); u0 = s.random_actions(50, B, seed=4)
            out = s.solve_device(x0, 50, u_init=u0); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5): out = s.solve_device(x0, 50, u_init=u0, workspace=out["workspace"])
            torch.cuda.synchronize()
            row.append(((time.perf_counter() - t) / 5 * 1e3, int(out["iterations"].max()) + 1))
        print(f"{name:24s} B={B:2d}: group {row[0][0]:7.2f} ms ({row[0][1]} it max)   wave {row[1][0]:7.2f} ms ({row[1][1]} it max)")
_hip.set_option("TFMPC_ILQR_KERNEL", None)
