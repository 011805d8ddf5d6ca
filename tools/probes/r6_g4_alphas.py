"""Four-wave groups, one vs two step sizes per wave and pass (TFMPC_GROUP4_ALPHAS; TFMPC_LIB = a variant build): shapes whose launch takes the four-wave form."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
def run(name, env, x0r, B, T=100):
    rng = np.random.default_rng(5)
    x0 = torch.as_tensor((np.array(x0r, dtype=np.float32)[None] * rng.uniform(0.9, 1.1, size=(B, 1, 1))).astype(np.float32), device="cuda")
    s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=1)
    out = None
    for _ in range(12): out = s.solve_device(x0, T, u_init=u0, workspace=None if out is None else out["workspace"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
    e1.record(); torch.cuda.synchronize()
    h = hashlib.sha256()
    for k in ("states", "actions", "costs", "iterations", "status"): h.update(out[k].cpu().numpy().tobytes())
    print(f"{name:28s} B {B:6d}: {e0.elapsed_time(e1) / 10:7.3f} ms  {h.hexdigest()[:12]}", flush=True)
run("res4 (512 groups)", Reservoir.load(dict(problems.RES4_CONFIG)), problems.RES4_X0, 32768)
run("hvac6 (512 groups)", HVAC.load(dict(problems.HVAC6_CONFIG)), problems.HVAC6_X0, 16384)
run("reservoir n=8 (512 groups)", Reservoir.load(dict(problems.reservoir_config(8, seed=3))), [[60.0]] * 8, 16384)
run("hvac n=12 (512 groups)", HVAC.load(dict(problems.hvac_config(12, seed=3))), [[12.0]] * 12, 8192)
run("reservoir n=16 (512 groups)", Reservoir.load(dict(problems.reservoir_config(16, seed=3))), [[60.0]] * 16, 8192)
