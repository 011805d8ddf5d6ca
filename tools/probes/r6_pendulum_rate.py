"""A torque-limited pendulum (n = 2, m = 1) written as DeviceEnv source -- tests/deviceenv_sources.py: PENDULUM -- B = 16 384, T = 50, <= 30 iterations:
ms per batch and iterations per second.  Round 6: 13.07 ms on the generic wave kernel -> 1.18 ms on the lane-group kernel (n + m <= 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch
import deviceenv_sources as t
from tfmpc.envs.deviceenv import DeviceEnv
from tfmpc.solvers.ilqr import iLQR
B, T = 16384, 50
rng = np.random.default_rng(5)
params = np.array([0.05, 9.81, 0.1, 0.0, 1.0, 0.1, 0.01], dtype=np.float32)
user = DeviceEnv(t.PENDULUM, 2, 1, params=params, low=-4.0, high=4.0)
x0 = np.stack([rng.uniform(-1.2, 1.2, size=B), rng.uniform(-1.0, 1.0, size=B)], axis=1).astype(np.float32)[..., None]
u0 = np.zeros((B, T, 1, 1), dtype=np.float32)
s = iLQR(user, max_iterations=30)
out = None
for _ in range(2): out = s.solve_device(x0, T, u_init=u0, workspace=None if out is None else out["workspace"])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
its = float((out["iterations"].double() + 1).sum())
print(f"pendulum B={B} T={T}: {ms:.2f} ms per batch, {its / B:.2f} iterations per instance, {its / ms / 1e3:.2f} M iterations/s, kernel {s.last_kernel}")
