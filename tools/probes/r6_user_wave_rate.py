"""A Python-defined HVAC with n = 12 (n + m = 24: thirty-two lanes per instance since round 6; before: the wave-per-instance costate kernel) and n = 20
(n + m = 40: the wave kernel) at B = 8 192, T = 100, <= 12 iterations: ms per solve."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems, torch_envs
from tfmpc.envs.hvac import HVAC
from tfmpc.solvers.ilqr import iLQR
for n in (12, 20):
    cfg = problems.hvac_config(n, seed=5)
    env = torch_envs.hvac(cfg, "cuda").to_device_env()
    B, T = 8192, 100
    rng = np.random.default_rng(3)
    x0 = rng.uniform(10.0, 30.0, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=12); u0 = iLQR(HVAC.load(cfg)).random_actions(T, B, seed=2)
    out = None
    for _ in range(2): out = s.solve_device(x0, T, u_init=u0, workspace=None if out is None else out["workspace"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
    e1.record(); torch.cuda.synchronize()
    print(f"hvac n={n} from Python: {e0.elapsed_time(e1) / 3:.2f} ms  kernel {s.last_kernel}  iterations {float((out['iterations'].double() + 1).mean()):.2f}  cost sum {float(out['costs'].double().sum()):.6e}", flush=True)
