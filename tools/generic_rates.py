"""Throughput of the shape-generic wave-per-instance kernels (fallback paths): LQR at shapes the matrix-core
kernel does not serve, and the dense iLQR kernels.  Run on the GPU box: python tools/generic_rates.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.solvers.lqr import LQR
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.lq import LQEnv
from tfmpc import _hip
from tfmpc.solvers.ilqr import iLQR

def t_lqr(n, m, B, T, force=None):
    _hip.set_option("TFMPC_LQR_KERNEL", force)
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1); F *= 0.5
    lqr = LQR(F, f, C, c); x0d = lqr._prep_x0(x0)
    out = lqr.solve_device(x0d, T); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): out = lqr.solve_device(x0d, T, workspace=out["workspace"])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    _hip.set_option("TFMPC_LQR_KERNEL", None)
    return dt

print(f"LQR generic n=16 m=8  T=50 B=8192: {t_lqr(16, 8, 8192, 50, 'generic')*1e3:.2f} ms")
print(f"LQR generic n=32 m=16 T=50 B=8192: {t_lqr(32, 16, 8192, 50)*1e3:.2f} ms")
print(f"LQR generic n=24 m=24 T=50 B=4096: {t_lqr(24, 24, 4096, 50)*1e3:.2f} ms")
_hip.set_option("TFMPC_ILQR_KERNEL", "wave")
env = Navigation.load(problems.NAV_CONFIG); s = iLQR(env, max_iterations=10)
rng = np.random.default_rng(4); B, T = 4096, 50
x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32); u0 = s.random_actions(T, B, seed=4)
out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
t = time.perf_counter(); out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
print(f"iLQR wave kernel, Navigation B=4096 T=50, <= 10 iterations: {(time.perf_counter()-t)*1e3:.1f} ms")
F, f, C, c, x0 = problems.make_lqr_batch_fast(2048, 16, 8, seed=1); F *= 0.25
s = iLQR(LQEnv(F, f, C, c), max_iterations=3); u0 = torch.zeros(2048, 50, 8, 1, device="cuda")
out = s.solve_device(x0[..., None].astype(np.float32), 50, u_init=u0); torch.cuda.synchronize()
t = time.perf_counter(); out = s.solve_device(x0[..., None].astype(np.float32), 50, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
print(f"iLQR wave kernel, LQ env n=16 m=8 B=2048 T=50, <= 3 iterations: {(time.perf_counter()-t)*1e3:.1f} ms")
