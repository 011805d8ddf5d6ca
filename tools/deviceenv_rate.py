"""Navigation (BASELINE configs[3]: n = m = 2, T = 50, B = 16 384) three ways: the built-in kernels (closed-form derivatives; lane-group kernel
and the generic wave kernel), the SAME env given as DeviceEnv source (tests/deviceenv_sources.py: wave kernel + dual numbers), and as torch
functions through TorchEnv (host-driven).  python tools/deviceenv_rate.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
import deviceenv_sources as sources
from tfmpc import _hip
from tfmpc.envs.deviceenv import DeviceEnv
from tfmpc.envs.navigation import Navigation
from tfmpc.solvers.ilqr import iLQR
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 50
cfg = problems.NAV_CONFIG
rng = np.random.default_rng(4)
x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
builtin = iLQR(Navigation.load(cfg))
u0 = builtin.random_actions(T, B, seed=4)
t0 = time.perf_counter()
user = iLQR(DeviceEnv(sources.NAVIGATION, 2, 2, params=sources.navigation_params(cfg), low=np.array(cfg["low"]), high=np.array(cfg["high"])))
user.env._library()
print(f"DeviceEnv ready (compile or cache hit) in {time.perf_counter() - t0:.2f} s")
def timed(solver, option=None):
    with _hip.option("TFMPC_ILQR_KERNEL", option):
        out = solver.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    its = float((out["iterations"].double() + 1).sum())
    timed.max_it = int(out["iterations"].max()) + 1
    return min(ts) * 1e3, its / min(ts), its / B
for name, solver, opt, wave in (("built-in env, lane-group kernel (default)", builtin, None, False), ("built-in env, generic wave kernel", builtin, "wave", False),
                                ("DeviceEnv source, lane-group kernel + dual numbers (default for n = m = 2)", user, None, False),
                                ("DeviceEnv source, generic wave kernel + dual numbers (any shape)", user, None, True)):
    user.env._library().force_wave_kernel(wave)
    ms, rate, mean_it = timed(solver, opt)
    print(f"{name}: {ms:.2f} ms per {B} solves, {rate / 1e6:.2f} M it/s, mean iterations {mean_it:.2f}, slowest instance {timed.max_it} iterations = {ms * 1e3 / timed.max_it:.1f} us each [{solver.last_kernel}]", flush=True)
user.env._library().force_wave_kernel(False)
