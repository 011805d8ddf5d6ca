"""cfg5 solve time against the batch size (n = m = 32, T = 100, 12 iterations): 16 384 instances are ONE wave per SIMD,
32 768 two.  python tools/cfg5_batch_sweep.py [B ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
n, T = int(os.environ.get("CFG5_N", "32")), 100
Bs = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 24576, 32768, 49152, 65536]
rng = np.random.default_rng(4)
for kind in ("hvac", "reservoir"):
    for B in Bs:
        if kind == "hvac":
            env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
        else:
            env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
        s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=5)
        with __import__("tfmpc")._hip.option("TFMPC_ILQR_KERNEL", "costate_mfma"):
            out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
        print(f"{kind:9s} B = {B:6d} ({B / 16 / 1024:.2f} waves per SIMD, n = {n}): {best * 1e3:7.2f} ms, {B * 12 / best / 1e6:6.2f} M iterations/s", flush=True)
        del out
