"""What the operand set-up (F~, C fragments from global memory, bounds-checked for padded shapes) costs an iLQR.solve on the LQ env:
the product kernel against a probe build that returns right after the set-up (a local edit: `return` before "start", loaded through TFMPC_LIB; measured 0.34 of 5.46 ms per 65 536 solves, round 3).
python tools/probes/prologue_cost.py   (run once per library)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
B, n, m, T = 65536, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=4321)
x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
u0 = torch.zeros(B, T, m, 1, device="cuda")
s = iLQR(LQEnv(0.25 * F, f, C, c))
out = s.solve_device(x0d, T, u_init=u0); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); out = s.solve_device(x0d, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"{os.environ.get('TFMPC_LIB', 'product')}: {min(ts):.3f} ms per {B} solves, iterations {float((out['iterations'].double() + 1).mean()):.2f}")
