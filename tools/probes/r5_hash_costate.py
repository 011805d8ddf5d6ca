"""sha256 of every output of a few costate-kernel solves (HVAC / Reservoir, one and two tiles, several group forms): run under two library builds
(TFMPC_LIB, tools/probes/r5_rev_ab.sh) to show that a change kept the bits."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
rng = np.random.default_rng(7)
for kind, n, B, T in (("hvac", 32, 4096, 100), ("hvac", 32, 300, 40), ("hvac", 6, 16384, 100), ("hvac", 6, 40000, 40), ("hvac", 16, 20000, 30), ("hvac", 9, 700, 25),
                      ("reservoir", 32, 4096, 100), ("reservoir", 4, 16384, 100), ("reservoir", 4, 40000, 40), ("reservoir", 8, 3000, 50), ("reservoir", 16, 600, 60)):
    if kind == "hvac":
        cfgd = dict(problems.HVAC6_CONFIG) if n == 6 else dict(problems.hvac_config(n, seed=5))
        env = HVAC.load(cfgd); x0 = rng.uniform(8.0, 25.0, size=(B, n, 1)).astype(np.float32)
    else:
        cfgd = dict(problems.RES4_CONFIG) if n == 4 else dict(problems.reservoir_config(n, seed=5))
        env = Reservoir.load(cfgd); x0 = rng.uniform(40.0, 80.0, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=6); u0 = s.random_actions(T, B, seed=n)
    out = s.solve_device(x0, T, u_init=u0, trace_rows=8); torch.cuda.synchronize()
    h = hashlib.sha256()
    for k in ("states", "actions", "costs", "iterations", "status", "trace_len"):
        h.update(out[k].cpu().numpy().tobytes())
    h.update(np.nan_to_num(out["trace"].cpu().numpy()).tobytes())
    print(f"{kind} n={n} B={B} T={T}: {h.hexdigest()[:16]}")
