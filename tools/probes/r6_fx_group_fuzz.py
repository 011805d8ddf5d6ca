"""Python-defined HVAC / Reservoir envs of every size n = 2 .. 16 (n + m = 4 .. 32) on the lane-packed costate kernel (sixteen / thirty-two lanes per
instance, csrc/user_env_group.h) against the wave-per-instance kernel of the same companion library: every output and the decision trace, bit for bit,
at random batch sizes and horizons.  One line per case; exit status 1 on any difference.
    python tools/probes/r6_fx_group_fuzz.py [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems, torch_envs
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for kind in ("hvac", "reservoir"):
    for n in range(3, 17):
        if kind == "hvac":
            cfg = problems.hvac_config(n, seed=int(rng.integers(1, 1000)))
            builtin, python_env, xr = HVAC.load(cfg), torch_envs.hvac(cfg, "cuda"), (5.0, 35.0)
        else:
            cfg = dict(problems.reservoir_config(n, seed=int(rng.integers(1, 1000))))
            builtin, python_env, xr = Reservoir.load(dict(cfg)), torch_envs.reservoir(cfg, "cuda"), (20.0, 95.0)
        env = python_env.to_device_env()
        assert env.zero_cost_hessian
        for rep in range(2):
            B, T, its = int(rng.integers(1, 400)), int(rng.integers(3, 70)), int(rng.integers(2, 10))
            x0 = rng.uniform(xr[0] + 0.2 * (xr[1] - xr[0]), xr[1] - 0.1 * (xr[1] - xr[0]), size=(B, n, 1)).astype(np.float32)
            solver = iLQR(env, max_iterations=its)
            u0 = iLQR(builtin).random_actions(T, B, seed=int(rng.integers(1, 1000)))
            outs = {}
            for wave in (False, True):
                env._library().force_wave_kernel(wave)
                try:
                    o = solver.solve_device(x0, T, u_init=u0, trace_rows=3 * its + 4)
                    torch.cuda.synchronize()
                    outs[wave] = ({k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}, solver.last_kernel)
                finally:
                    env._library().force_wave_kernel(False)
            same = all(torch.equal(outs[False][0][k], outs[True][0][k]) for k in ("states", "actions", "costs", "iterations", "status", "trace_len"))
            same = same and torch.equal(torch.nan_to_num(outs[False][0]["trace"], nan=-7.0), torch.nan_to_num(outs[True][0]["trace"], nan=-7.0))
            moved = len(torch.unique(outs[False][0]["trace"][:, :, 5].nan_to_num(nan=-1.0)))
            bad += 0 if same else 1
            print(f"{kind:9s} n={n:2d} B={B:3d} T={T:2d} iterations<={its}  {outs[False][1][:41]:41s} vs {outs[True][1][:4]}  step sizes seen {moved:2d}  "
                  f"{'same bits' if same else 'DIFFERENT'}", flush=True)
print("cases with a difference:", bad)
sys.exit(1 if bad else 0)
