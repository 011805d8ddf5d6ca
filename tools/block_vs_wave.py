"""Workgroup-per-instance (lqr_block.hip) vs wave-per-instance (lqr_generic.hip) LQR kernels over shapes and batch
sizes: where the dispatcher's threshold (kBlockFrom in lqr_dispatch.hip) should sit.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.solvers.lqr import LQR


def t_lqr(n, m, B, T, force):
    _hip.set_option("TFMPC_LQR_KERNEL", force)
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1); F *= 1.0 / np.sqrt(n)
    lqr = LQR(F, f, C, c); x0d = lqr._prep_x0(x0)
    out = lqr.solve_device(x0d, T); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): out = lqr.solve_device(x0d, T, workspace=out["workspace"])
    torch.cuda.synchronize()
    _hip.set_option("TFMPC_LQR_KERNEL", None)
    return (time.perf_counter() - t) / 5 * 1e3


shapes = [(9, 9), (10, 10), (17, 2), (18, 4), (20, 4), (24, 8), (24, 12), (32, 16), (32, 32), (48, 16)]
for B in (256, 8192):
    for n, m in shapes:
        g, k = t_lqr(n, m, B, 50, "generic"), t_lqr(n, m, B, 50, "block")
        print(f"B={B:5d} n={n:2d} m={m:2d}: wave {g:8.2f} ms   block {k:8.2f} ms   ratio {g / k:5.2f}")
