"""The headline shape through iLQR.solve on the LQ env (n = 16, m = 8, T = 50, B = 65 536) -- bench.py's `extra.ilqr_api`
workload: make_lqr's distribution (tests/problems.py:make_lqr_batch_spd) with F scaled by 0.25, start = LQR-optimal
actions + 5 % noise; unbounded (argument "box": actions in [-0.5, 0.5], the control-limited kernel): one warm-up + 2
launches, for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
from tfmpc.solvers.lqr import LQR
box = len(sys.argv) > 1 and sys.argv[1] == "box"
B, n, m, T = 65536, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=4321)
F = 0.25 * F
x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
opt = LQR(F, f, C, c).solve_device(x0d, T)["actions"]
gen = torch.Generator(device="cuda").manual_seed(7)
u0 = (opt + 0.05 * opt.abs().amax(dim=(1, 2, 3), keepdim=True) * torch.randn(opt.shape, device="cuda", generator=gen)).contiguous()
env = LQEnv(F, f, C, c, low=-0.5, high=0.5) if box else LQEnv(F, f, C, c)
s = iLQR(env)
out = s.solve_device(x0d, T, u_init=u0)
for _ in range(2): out = s.solve_device(x0d, T, u_init=u0, workspace=out["workspace"])
torch.cuda.synchronize()
print("iterations", float((out["iterations"].double() + 1).sum()), "flagged", int((out["status"] != 0).sum()))
