"""Line-search trace of the cfg5 kernel (probe build: tools/probes/build_probe.sh ilqr_adjoint_mfma.hip, loaded through
TFMPC_LIB): per instance and sweep the accepted step-size index and, per step size tried, the first time step at which the
partial cost exceeded J_hat.  Writes gpurun_out/cfg5_trace_<env>.npy ([B][16][12] int32) for the offline regrouping model
(tools/probes/cfg5_regroup_model.py)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TFMPC_LIB", os.path.join(ROOT, "tools/probes/ab/lib_probe.so"))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
n, T, B = 32, 100, 32768
lib = _hip.load()
lib.tfmpc_debug_cfg5_trace.argtypes = [ctypes.c_void_p]
rng = np.random.default_rng(4)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for kind in ("hvac", "reservoir"):
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=5)
    trace = torch.full((B, 16, 12), -1, dtype=torch.int32, device="cuda")
    assert lib.tfmpc_debug_cfg5_trace(trace.data_ptr()) == 0
    out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    np.save(os.path.join(ROOT, f"gpurun_out/cfg5_trace_{kind}.npy"), trace.cpu().numpy().astype(np.int16))
    print(kind, "traced; mean iterations", float((out["iterations"].float() + 1).mean()))
