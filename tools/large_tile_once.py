"""One launch each of the two large-tile matrix-core kernels, for profiling (rocprofv3 ... -- python3 tools/large_tile_once.py):
lqr_mfma32x16_kernel (dense LQR n = 32, m = 16, T = 50, B = 8192) and ilqr_lq_mfma32_kernel (iLQR on the LQ env at BASELINE
configs[4]'s literal dims n = 32, m = 16, T = 100, B = 8192)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
from tfmpc.solvers.lqr import LQR
n, m, B = 32, 16, 8192
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1)
lqr = LQR(F * 0.5, f, C, c); x0d = lqr._prep_x0(x0)
out = lqr.solve_device(x0d, 50); torch.cuda.synchronize()
t = time.perf_counter(); out = lqr.solve_device(x0d, 50, workspace=out["workspace"]); torch.cuda.synchronize()
print(f"lqr_mfma32x16: {(time.perf_counter() - t) * 1e3:.2f} ms for {B} solves")
s = iLQR(LQEnv(F * (0.9 / np.sqrt(n)), f, C, c))
xd = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda"); u0 = torch.zeros(B, 100, m, 1, device="cuda")
out = s.solve_device(xd, 100, u_init=u0); torch.cuda.synchronize()
t = time.perf_counter(); out = s.solve_device(xd, 100, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
print(f"ilqr_lq_mfma32: {(time.perf_counter() - t) * 1e3:.2f} ms for {B} solves, {float((out['iterations'].double() + 1).mean()):.2f} iterations each")
