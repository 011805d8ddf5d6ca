import os, sys
ROOT = "/root/repo"
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems, torch_envs
from oracle import envs_ref, ilqr_ref
from tfmpc import _hip
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
cfg = dict(problems.RES4_CONFIG)
builtin = Reservoir.load(dict(cfg)); dev = torch_envs.reservoir(cfg, "cuda").to_device_env()
rng = np.random.default_rng(8); B, T, n = 24, 30, 4
x0 = rng.uniform(20 + 0.3 * 75, 95 - 0.2 * 75, size=(B, n, 1)).astype(np.float32)
u0 = iLQR(builtin).random_actions(T, B, seed=2)
a = iLQR(dev, max_iterations=1).solve_device(x0, T, u_init=u0)
with _hip.option("TFMPC_ILQR_KERNEL", "wave"):
    b = iLQR(builtin, max_iterations=1).solve_device(x0, T, u_init=u0)
c = iLQR(builtin, max_iterations=1).solve_device(x0, T, u_init=u0)
torch.cuda.synchronize()
o = ilqr_ref.ILQRRef(envs_ref.Reservoir(**cfg), max_iterations=1)
xs, us, cs, it = o.solve(x0[0].astype(np.float64), T, u_init=u0[0].cpu().numpy().astype(np.float64))
print("iters", int(a["iterations"][0]), int(b["iterations"][0]), int(c["iterations"][0]), it)
print("dev    last costs", a["costs"][0, -3:].tolist(), "last u", a["actions"][0, -1].flatten().tolist())
print("wave   last costs", b["costs"][0, -3:].tolist(), "last u", b["actions"][0, -1].flatten().tolist())
print("dflt   last costs", c["costs"][0, -3:].tolist(), "last u", c["actions"][0, -1].flatten().tolist())
print("oracle last costs", cs[-3:].tolist(), "last u", us[-1].flatten().tolist())
print("x_T dev", a["states"][0, -1].flatten().tolist()); print("x_T orc", xs[-1].flatten().tolist())
