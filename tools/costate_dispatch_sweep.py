"""Where should the dispatcher hand a shared-env HVAC / Reservoir batch at n > 16 to the 16-per-wave matrix-core kernel
instead of the register-resident one-instance-per-wave kernel?  Times both (forced) over small batches.  n = 32, T = 100."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
n, T = int(os.environ.get("CFG5_N", "32")), 100
rng = np.random.default_rng(4)
for kind in ("hvac", "reservoir"):
    for B in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 3072, 4096]:
        if kind == "hvac":
            env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
        else:
            env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
        s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=5)
        res = {}
        for kern in ("lean", "costate_mfma"):
            with _hip.option("TFMPC_ILQR_KERNEL", kern):
                out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter(); out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
            res[kern] = best * 1e3
        print(f"{kind:9s} n = {n} B = {B:5d}: register-resident {res['lean']:6.2f} ms, 16 per wave {res['costate_mfma']:6.2f} ms", flush=True)
