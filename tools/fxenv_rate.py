"""Navigation (configs[3]: B = 16 384, T = 50) from PLAIN TORCH FUNCTIONS through TorchEnv.to_device_env() beside the hand-written DeviceEnv
source and the host-driven TorchEnv solve (on a sample): python tools/fxenv_rate.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch
sys.argv = sys.argv[:1]
import bench
r = bench.deviceenv_rate()
print(json.dumps({"hand_written_Mit_s": r["iterations_per_s"] / 1e6, "hand_written_ms": r["ms_per_batch"], "from_python": r["from_python_functions"],
                  "builtin_lane_group_ms": r["same_env_builtin_lane_group_kernel_ms"]}, indent=1))

# ... and the reference's own res4 / hvac6 configs from Python (B = 16 384, T = 100, <= 12 iterations) beside the built-in env on the SAME generic wave
# kernel (TFMPC_ILQR_KERNEL=wave) and on its specialised 16-instances-per-wave kernel (default): what a user-defined env of that size can expect
import problems, torch_envs, workloads
from tfmpc import _hip
from tfmpc.solvers.ilqr import iLQR
for name, make, cfg in (("res4", torch_envs.reservoir, problems.RES4_CONFIG), ("hvac6", torch_envs.hvac, problems.HVAC6_CONFIG)):
    w = workloads.small_env(name)
    x0, u0, T = w["x0"], w["u0"], w["T"]
    def timed(solver, option=None):
        with _hip.option("TFMPC_ILQR_KERNEL", option):
            out = solver.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 3 * 1e3, float((out["iterations"].double() + 1).sum())
    t0 = time.perf_counter()
    py = iLQR(make(dict(cfg), "cuda").to_device_env(), max_iterations=12)
    py.env._library()
    ready = time.perf_counter() - t0
    ms_py, its_py = timed(py)
    ms_wave, its_w = timed(w["solver"], "wave")
    ms_def, its_d = timed(w["solver"])
    print(json.dumps({name: {"from_python_ms": ms_py, "from_python_Mit_s": its_py / ms_py / 1e3, "trace_translate_compile_s": ready,
                             "builtin_env_generic_wave_kernel_ms": ms_wave, "builtin_env_specialised_kernel_ms": ms_def,
                             "kernel": py.last_kernel}}))
