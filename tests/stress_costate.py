"""Opt-in randomised check (not collected by pytest): register-resident HVAC / Reservoir kernel == generic wave
kernels (one instance per wave, or several for small n) == generic wave kernel, bit for bit, over random sizes 2..32,
horizons, batch sizes, seeds and iteration caps.
Run on the GPU box: python tests/stress_costate.py [cases]"""
import os, sys
sys.path.insert(0, '/root/repo/tf-mpc_amd'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(7)
bad = 0
for case in range(cases):
    kind = "hvac" if case % 2 == 0 else "reservoir"
    n = int(rng.integers(2, 33)); T = int(rng.integers(1, 60)); B = int(rng.integers(1, 300)); mi = int(rng.integers(1, 16))
    seed = int(rng.integers(0, 10_000))
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=seed))); x0 = rng.uniform(0.0, 40.0, size=(B, n, 1)).astype(np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=seed))); x0 = rng.uniform(5.0, 99.0, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=mi)
    u0 = s.random_actions(T, B, seed=seed)
    out = {}
    for kern in (None, "wave"):
        if kern is None: os.environ.pop("TFMPC_ILQR_KERNEL", None)
        else: os.environ["TFMPC_ILQR_KERNEL"] = kern
        out[kern] = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    os.environ.pop("TFMPC_ILQR_KERNEL", None)
    same = all(torch.equal(out[None][k], out["wave"][k]) for k in ("states", "actions", "costs", "iterations", "status"))
    finite = bool(torch.isfinite(out[None]["costs"]).all())
    bad += not same
    print(f"case {case:3d} {kind:9s} n={n} T={T:2d} B={B:3d} max_iterations={mi}: identical={same} finite={finite} "
          f"mean iterations {float((out[None]['iterations'].float() + 1).mean()):.1f}", flush=True)
print("MISMATCHES:", bad)
