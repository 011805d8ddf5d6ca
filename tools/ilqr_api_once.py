"""The headline shape through iLQR.solve on the LQ env (n = 16, m = 8, T = 50, B = 65 536), unbounded (argument "box":
actions in [-0.5, 0.5], the control-limited kernel): one warm-up + 2 launches, for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
box = len(sys.argv) > 1 and sys.argv[1] == "box"
B, n, m, T = 65536, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=4321)
env = LQEnv(0.25 * F, f, C, c, low=-0.5, high=0.5) if box else LQEnv(0.25 * F, f, C, c)
s = iLQR(env)
x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda"); u0 = torch.zeros(B, T, m, 1, device="cuda")
out = s.solve_device(x0d, T, u_init=u0)
for _ in range(2): out = s.solve_device(x0d, T, u_init=u0, workspace=out["workspace"])
torch.cuda.synchronize()
print("iterations", float((out["iterations"].double() + 1).sum()))
