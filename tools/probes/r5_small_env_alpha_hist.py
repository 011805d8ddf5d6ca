"""Which step size do the line searches of the small-env workloads (tools/small_env_rates.py) accept, pass by pass?  From the device trace:
histogram of the accepted position (`alpha_index`) per iteration, and how often a column's position moves up / down between iterations.
Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR, TRACE_COLUMNS

rng = np.random.default_rng(0)
for name, env, x0 in (("hvac6", HVAC.load(dict(problems.HVAC6_CONFIG)), np.array(problems.HVAC6_X0, dtype=np.float32)),
                      ("res4", Reservoir.load(dict(problems.RES4_CONFIG)), np.array(problems.RES4_X0, dtype=np.float32))):
    B, T = 16384, 100
    x = (x0[None] * rng.uniform(0.9, 1.1, size=(B, 1, 1))).astype(np.float32)
    s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=1)
    out = s.solve_device(x, T, u_init=u0, trace_rows=16)
    tr = out["trace"].cpu().numpy(); ln = out["trace_len"].cpu().numpy()
    ia, iacc = TRACE_COLUMNS.index("alpha_index"), TRACE_COLUMNS.index("accepted")
    print(name, "kernel", s.last_kernel, "passes per instance", ln.mean())
    for p in range(int(ln.max())):
        a = tr[:, p, ia]; ok = np.isfinite(a)
        h = np.bincount(a[ok].astype(int), minlength=11)
        print(f"  pass {p:2d}: accepted {np.nanmean(tr[:, p, iacc]):.3f}  position histogram {h.tolist()}")
    per_group = 64 if name == "res4" else 32
    a = tr[:, :12, ia]
    g = a[: (B // per_group) * per_group].reshape(-1, per_group, 12)
    gmax, gmin = np.nanmax(g, axis=1), np.nanmin(g, axis=1)          # per group and pass
    print("  per group: mean max position", np.nanmean(gmax, axis=0).round(2).tolist())
    print("  per group: mean min position", np.nanmean(gmin, axis=0).round(2).tolist())
    up = (gmax[:, 1:] > gmax[:, :-1] + 0).mean(axis=0)
    print("  share of groups whose max position rises against the pass before:", up.round(3).tolist())
    up1 = (gmax[:, 1:] > gmax[:, :-1] + 1).mean(axis=0)
    print("  ... by more than one:", up1.round(3).tolist())
