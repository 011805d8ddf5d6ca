"""A/B: res4 (the reference's own Reservoir config, n = 4, four instances per matrix-core column) and a chain of 12 reservoirs (one per
column) with the shift fast path (default) against TFMPC_COSTATE_COUPLING=dense, alternating launches on one box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
rng = np.random.default_rng(0)
for name, env, x0v in (("res4", Reservoir.load(dict(problems.RES4_CONFIG)), np.array(problems.RES4_X0, dtype=np.float32)),
                       ("chain of 12", Reservoir.load(dict(problems.reservoir_config(12, seed=5))), np.full((12, 1), 60.0, dtype=np.float32))):
    for B in (16384, 65536):
        T = 100
        x = (x0v[None] * rng.uniform(0.9, 1.1, size=(B, 1, 1))).astype(np.float32)
        s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=1)
        out = s.solve_device(x, T, u_init=u0); torch.cuda.synchronize()
        ts, ref = {None: [], "dense": []}, None
        for rep in range(6):
            for mode in (None, "dense"):
                with _hip.option("TFMPC_COSTATE_COUPLING", mode):
                    t0 = time.perf_counter(); out = s.solve_device(x, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
                    ts[mode].append((time.perf_counter() - t0) * 1e3)
                cur = out["states"].clone()
                if ref is None: ref = cur
                assert torch.equal(ref, cur)
        print(f"{name} B={B}: shift {min(ts[None]):.3f} ms (median {np.median(ts[None]):.3f}) | dense {min(ts['dense']):.3f} ms (median {np.median(ts['dense']):.3f}); same bits")
