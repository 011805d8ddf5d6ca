"""iLQR.solve on the LQ env, headline shape, 65 536 problems (bench.py's ilqr_api workload): time and outputs of the loaded build;
python tools/probes/api_ab.py save|compare <file.npz>   (TFMPC_LIB selects the build per process)"""
import os, sys, time
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
s = iLQR(LQEnv(F, f, C, c))
out = s.solve_device(x0d, T, u_init=u0); torch.cuda.synchronize()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); out = s.solve_device(x0d, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"{min(ts):.3f} ms per {B} solves (runs {' '.join(f'{t:.2f}' for t in ts)}), iterations {float((out['iterations'].double() + 1).mean()):.3f}")
res = {k: out[k].cpu().numpy() for k in ("states", "actions", "costs", "iterations", "status")}
if sys.argv[1] == "save":
    np.savez(sys.argv[2], **res)
else:
    ref = np.load(sys.argv[2])
    print("bit-identical:", {k: bool(np.array_equal(res[k], ref[k])) for k in res})
