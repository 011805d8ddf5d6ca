"""Replays one case of tools/probes/fuzz_costate.py with details.  python tools/probes/fuzz_case.py <case>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
target = int(sys.argv[1])
rng = np.random.default_rng(2026)
for case in range(target + 1):
    kind = "reservoir" if case % 3 else "hvac"
    n, T, B = int(rng.integers(1, 33)), int(rng.integers(1, 60)), int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 200, 1000, 5000]))
    its = int(rng.integers(1, 9))
    if kind == "hvac":
        x0 = rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32); its = 1
    else:
        x0 = rng.uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)
    atol = float(rng.choice([5e-3, 0.05]))
env = (HVAC.load(dict(problems.hvac_config(n, seed=case))) if kind == "hvac" else Reservoir.load(dict(problems.reservoir_config(n, seed=case))))
print(kind, n, T, B, its, atol)
for max_it in range(1, its + 1):
    s = iLQR(env, max_iterations=max_it, atol=atol)
    u0 = s.random_actions(T, B, seed=case)
    out = {}
    for kern in ("wave", "lean", "costate_mfma"):
        with _hip.option("TFMPC_ILQR_KERNEL", kern):
            o = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
            out[kern] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}
    w = out["wave"]
    for kern in ("lean", "costate_mfma"):
        f = out[kern]
        print(f" max_iterations {max_it} {kern:13s}:", {k: (bool(torch.equal(w[k], f[k])), float((w[k].double() - f[k].double()).abs().max())) for k in ("states", "actions", "costs", "iterations", "status")},
              "iterations", w["iterations"].tolist(), f["iterations"].tolist(), "status", w["status"].tolist(), f["status"].tolist())
