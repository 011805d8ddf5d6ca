"""Dense LQR beyond the 16 x 8 tile: the 2 x 2-tile matrix-core kernel (lqr_mfma32x16.hip, default) against the
workgroup-per-instance f32-MFMA kernel (TFMPC_LQR_KERNEL=block), T = 50.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.solvers.lqr import LQR

def t_lqr(n, m, B, T, force):
    _hip.set_option("TFMPC_LQR_KERNEL", force)
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1); F *= 0.5
    lqr = LQR(F, f, C, c); x0d = lqr._prep_x0(x0)
    out = lqr.solve_device(x0d, T); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): out = lqr.solve_device(x0d, T, workspace=out["workspace"])
    torch.cuda.synchronize()
    _hip.set_option("TFMPC_LQR_KERNEL", None)
    return (time.perf_counter() - t) / 5 * 1e3, int((out["status"] != 0).sum())

for (n, m) in ((32, 16), (24, 12), (20, 4)):
    for B in (256, 2048, 8192, 32768):
        a, fa = t_lqr(n, m, B, 50, None); b, fb = t_lqr(n, m, B, 50, "block")
        print(f"n={n} m={m} B={B:6d}: mfma_32x16 {a:8.3f} ms ({B/a/1e3:8.1f} k solves/s, flagged {fa})   block {b:8.3f} ms   x{b/a:.2f}", flush=True)
