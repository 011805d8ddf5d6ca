"""Does a result depend on what the workspace held before?  Reservoir / HVAC through the 16-per-wave kernel with the
workspace pre-filled with NaN, against a zero-filled one, on batches whose last matrix-core column is only partly live."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
rng = np.random.default_rng(1)
bad = 0
for kind, n, B in (("reservoir", 2, 2), ("reservoir", 4, 5), ("reservoir", 7, 3), ("hvac", 3, 2), ("hvac", 6, 33), ("reservoir", 20, 5), ("hvac", 32, 17)):
    T = 30
    env = HVAC.load(dict(problems.hvac_config(n, seed=1))) if kind == "hvac" else Reservoir.load(dict(problems.reservoir_config(n, seed=1)))
    x0 = rng.uniform(20.0, 60.0, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=5); u0 = s.random_actions(T, B, seed=3)
    with _hip.option("TFMPC_ILQR_KERNEL", "costate_mfma"):
        ref = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
        ws = ref["workspace"]
        outs = []
        for fill in (0.0, float("nan"), float("inf")):
            ws.fill_(fill)
            o = s.solve_device(x0, T, u_init=u0, workspace=ws); torch.cuda.synchronize()
            outs.append({k: o[k].clone() for k in ("states", "actions", "costs", "iterations", "status")})
    same = all(torch.equal(outs[0][k], o[k]) or (torch.isnan(outs[0][k]) == torch.isnan(o[k])).all() and torch.equal(torch.nan_to_num(outs[0][k]), torch.nan_to_num(o[k])) for o in outs[1:] for k in outs[0])
    bad += not same
    print(f"{kind} n={n} B={B}: {'independent of the workspace contents' if same else 'DEPENDS on the workspace contents'}", flush=True)
sys.exit(1 if bad else 0)
