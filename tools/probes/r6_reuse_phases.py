"""Round 6: where the time of the gain-reusing LQ kernel goes -- the bench's `ilqr_api` workload with the iteration count capped at 1, 2, 3
(an instance stops at its cap), with TFMPC_ILQR_LQ_REUSE on and off."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
from tfmpc.solvers.lqr import LQR
B, n, m, T = int(os.environ.get("B", 65536)), 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=4321)
F = 0.25 * F
x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
opt = LQR(F, f, C, c).solve_device(x0d, T)["actions"]
gen = torch.Generator(device="cuda").manual_seed(7)
u0 = (opt + 0.05 * opt.abs().amax(dim=(1, 2, 3), keepdim=True) * torch.randn(opt.shape, device="cuda", generator=gen)).contiguous()
env = LQEnv(F, f, C, c)
res = {}
for cap in (1, 2, 3, 100):
    for mode in (None, "0"):
        s = iLQR(env, max_iterations=cap)
        with _hip.option("TFMPC_ILQR_LQ_REUSE", mode):
            o = s.solve_device(x0d, T, u_init=u0)
            for _ in range(3): o = s.solve_device(x0d, T, u_init=u0, workspace=o["workspace"])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): o = s.solve_device(x0d, T, u_init=u0, workspace=o["workspace"])
            e1.record(); torch.cuda.synchronize()
        res[f"cap{cap}_reuse{'on' if mode is None else 'off'}"] = [round(e0.elapsed_time(e1) / 10, 4), float((o["iterations"].double() + 1).sum()), int((o["status"] != 0).sum())]
print(json.dumps(res, indent=1))
