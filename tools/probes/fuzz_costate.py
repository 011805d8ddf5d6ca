"""One-off fuzz of the 16-per-wave costate kernel against the wave kernel on random shapes: Reservoir must agree bit for bit
(states, actions, costs, iterations, status), HVAC to 1e-5 of the state scale on the first iteration.  python tools/probes/fuzz_costate.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2026)
bad = 0
for case in range(cases):
    kind = "reservoir" if case % 3 else "hvac"
    n, T, B = int(rng.integers(1, 33)), int(rng.integers(1, 60)), int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 200, 1000, 5000]))
    its = int(rng.integers(1, 9))
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=case))); x0 = rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32)
        its = 1
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=case))); x0 = rng.uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=its, atol=float(rng.choice([5e-3, 0.05])))
    u0 = s.random_actions(T, B, seed=case)
    out = {}
    for kern in ("wave", "costate_mfma"):
        with _hip.option("TFMPC_ILQR_KERNEL", kern):
            o = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
            out[kern] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}
    w, f = out["wave"], out["costate_mfma"]
    if kind == "reservoir":
        ok = all(torch.equal(w[k], f[k]) for k in ("states", "actions", "costs", "iterations", "status"))
    else:
        scale = float(w["states"].abs().max())
        ok = float((w["states"] - f["states"]).abs().max()) <= 1e-5 * scale and torch.equal(w["status"], f["status"])
    bad += not ok
    print(f"case {case:3d} {kind:9s} n={n:2d} T={T:2d} B={B:5d} iterations<={its}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
