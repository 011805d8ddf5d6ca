"""A/B: the two-tile Reservoir kernel with the shift fast path (default) against TFMPC_COSTATE_COUPLING=dense (the bf16x3 products),
cfg5 (n = 32, T = 100, B = 32 768, 12 iterations) and B = 8 192; alternating launches on one box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
n, T = 32, 100
for B in (32768, 8192):
    rng = np.random.default_rng(4)
    env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=5)
    out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    ts = {None: [], "dense": []}
    ref = None
    for rep in range(6):
        for mode in (None, "dense"):
            with _hip.option("TFMPC_COSTATE_COUPLING", mode):
                t0 = time.perf_counter(); out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
                ts[mode].append((time.perf_counter() - t0) * 1e3)
            cur = (out["states"].clone(), out["iterations"].clone())
            if ref is None: ref = cur
            assert torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1])
    print(f"B={B}: shift {min(ts[None]):.2f} ms (median {np.median(ts[None]):.2f}) | dense products {min(ts['dense']):.2f} ms (median {np.median(ts['dense']):.2f}); same bits")
