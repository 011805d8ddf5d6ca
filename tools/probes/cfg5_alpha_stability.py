"""How predictable is the step size a column accepts?  Traced cfg5 solves (n = 32, T = 100, 12 iterations, 2 048 instances):
histogram of the accepted position in the 11-point list and P(position == the position accepted in the previous iteration).
Decides whether a line-search pass should WRITE the candidate of the predicted step size (saving the stored re-rollout)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
n, T, B = 32, 100, 2048
rng = np.random.default_rng(4)
for kind in ("hvac", "reservoir"):
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=5)
    out = s.solve_device(x0, T, u_init=u0, trace_rows=16); torch.cuda.synchronize()
    tr = out["trace"].cpu().numpy(); ln = out["trace_len"].cpu().numpy()
    hist = np.zeros(12, dtype=np.int64); same = later = earlier = total = 0; by_iter = {}
    for b in range(B):
        prev = None
        for r in tr[b, :min(int(ln[b]), tr.shape[1])]:
            if r[8] <= 0: continue                       # converged on g_norm, or rejected
            idx = int(r[5]); hist[idx] += 1
            if prev is not None:
                total += 1; same += idx == prev; later += idx > prev; earlier += idx < prev
                d = by_iter.setdefault(int(r[0]), [0, 0]); d[0] += idx == prev; d[1] += 1
            prev = idx
    print(kind, "accepted position histogram", hist.tolist())
    print(kind, f"same as previous iteration {same / max(total, 1):.3f}, later {later / max(total, 1):.3f}, earlier {earlier / max(total, 1):.3f}")
    print(kind, "hit rate by iteration", {k: round(v[0] / v[1], 2) for k, v in sorted(by_iter.items())})
