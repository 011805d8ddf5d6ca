"""Solve time of the 16-per-wave costate kernel per group form (TFMPC_COSTATE_WAVES = 1 | 2 | 4 | 8) over batch sizes:
python tools/probes/group_waves_sweep.py <hvac|reservoir|hvac6|res4> <n> <B,B,...> [waves,waves,...] [kernel]
(kernel: costate_mfma forces the 16-per-wave kernel where the dispatch would take the register-resident one; "default")"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
kind, n = sys.argv[1], int(sys.argv[2])
Bs = [int(v) for v in sys.argv[3].split(",")]
waves = sys.argv[4].split(",") if len(sys.argv) > 4 else ["1", "2", "4", "8"]
kernel = sys.argv[5] if len(sys.argv) > 5 else "default"
T = 100
rng = np.random.default_rng(4)
for B in Bs:
    if kind == "hvac6":
        env = HVAC.load(dict(problems.HVAC6_CONFIG)); x0 = np.tile(np.array(problems.HVAC6_X0, dtype=np.float32)[None], (B, 1, 1))
    elif kind == "res4":
        env = Reservoir.load(dict(problems.RES4_CONFIG)); x0 = np.tile(np.array(problems.RES4_X0, dtype=np.float32)[None], (B, 1, 1))
    elif kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=5)
    row = []
    for w in waves:
        with _hip.option("TFMPC_COSTATE_WAVES", None if w == "auto" else w), _hip.option("TFMPC_ILQR_KERNEL", None if kernel == "default" else kernel):
            out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
        row.append(f"{w}: {min(ts):6.2f}")
    print(f"{kind} n={env.state_size} B={B:6d} kernel={kernel}  ms by waves  " + "  ".join(row), flush=True)
