"""One HVAC and one Reservoir cfg5 solve (n = m = 32, T = 100, B = 32768, 12 iterations) for profiling:
rocprofv3 --kernel-trace --pmc ... -- python3 tools/cfg5_once.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
n, T, B = 32, 100, int(os.environ.get("CFG5_B", 32768))
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"          # 16-bit trajectory containers (storage_bf16)
rng = np.random.default_rng(4)
for kind in ("hvac", "reservoir"):
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=12, storage_bf16=bf16); u0 = s.random_actions(T, B, seed=5)
    out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    import time
    ts = []
    for _ in range(1 if os.environ.get("ROCPROFILER_REGISTER_ROOT") or os.environ.get("CFG5_ONCE_SINGLE") else 5):
        t0 = time.perf_counter(); out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{kind} {'bf16' if bf16 else 'fp32'} containers: {min(ts):.2f} ms (runs: {' '.join(f'{t:.2f}' for t in ts)})")
    print(kind, "mean iterations", float((out["iterations"].float() + 1).mean()), "mean total cost", float(out["costs"].sum(dim=1).mean()))
