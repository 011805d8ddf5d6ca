"""The LQ env (n = 16, m = 8, T = 50) written as DeviceEnv source against the built-in LQ env on the generic wave kernel and on its matrix-core kernel:
what a DENSE user env of that size costs (dual numbers: 24 directions of `transition`, 300 second-order pairs of `cost` per time step).
    python tools/deviceenv_lq_rate.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
import deviceenv_sources as sources
from tfmpc import _hip
from tfmpc.envs.deviceenv import DeviceEnv
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
B, n, m, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=3)
F = 0.25 * F
x0 = x0.astype(np.float32)[..., None]
u0 = torch.zeros(B, T, m, 1, device="cuda")
builtin = iLQR(LQEnv(F, f, C, c))
user = iLQR(DeviceEnv(sources.lq_source(n, m), n, m, params=sources.lq_params(F, f, C, c)))
def timed(solver, option=None):
    with _hip.option("TFMPC_ILQR_KERNEL", option):
        out = solver.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    its = float((out["iterations"].double() + 1).sum())
    return dt * 1e3, its / dt, its / B
for name, solver, opt in (("built-in LQ env, matrix-core kernel (default)", builtin, None), ("built-in LQ env, generic wave kernel", builtin, "wave"),
                          ("LQ env as DeviceEnv source, generic wave kernel + dual numbers", user, None)):
    ms, rate, mean_it = timed(solver, opt)
    print(f"{name}: {ms:.2f} ms per {B} solves, {rate / 1e3:.1f} k it/s, mean iterations {mean_it:.2f} [{solver.last_kernel}]", flush=True)
