"""Device time of bench.py's `ilqr_api` workload (iLQR.solve on the LQ env, n = 16, m = 8, T = 50, B = 65 536; tools/ilqr_api_once.py), warm start, unbounded and
(argument "box") control-limited on the stable variant: median of 7 launches between two events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
from tfmpc.solvers.lqr import LQR
B, n, m, T = 65536, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=4321)
F = 0.25 * F
x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
opt = LQR(F, f, C, c).solve_device(x0d, T)["actions"]
gen = torch.Generator(device="cuda").manual_seed(7)
u0 = (opt + 0.05 * opt.abs().amax(dim=(1, 2, 3), keepdim=True) * torch.randn(opt.shape, device="cuda", generator=gen)).contiguous()
for name, env in (("unbounded", LQEnv(F, f, C, c)), ("limited +-0.5", LQEnv(F, f, C, c, low=-0.5, high=0.5))):
    s = iLQR(env)
    out = s.solve_device(x0d, T, u_init=u0); torch.cuda.synchronize()
    ts = []
    for _ in range(7 if name == "unbounded" else 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = s.solve_device(x0d, T, u_init=u0, workspace=out["workspace"]); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    import hashlib
    h = hashlib.sha256(out["states"].cpu().numpy().tobytes() + out["actions"].cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"ilqr_api {name}: {sorted(ts)[len(ts) // 2]:.3f} ms (min {min(ts):.3f}), iterations {float((out['iterations'].double() + 1).mean()):.2f}, sha {h}")
