"""Can the stragglers of the stable control-limited workload be told in advance?  Per instance: the work it ends up doing (sweeps + 0.4 line-search
rollouts, from the decision trace) against what a cheap pre-pass could know -- the start cost J_0, the first pass's regularisation level,
gradient norm, accepted step size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
from tfmpc.solvers.ilqr import TRACE_COLUMNS
w = workloads.control_limited_stable(65536)
out = workloads.solver_of(w).solve_device(w["x0"], w["T"], u_init=w["u0"], trace_rows=170)
torch.cuda.synchronize()
rows = torch.nan_to_num(out["trace"], nan=-1.0)
valid = torch.arange(rows.shape[1], device=rows.device)[None, :] < out["trace_len"][:, None]
searched = valid & (rows[..., 8] >= 0)
roll = ((rows[..., 5].clamp(min=0) + 1) * searched).sum(1).cpu().numpy()
sweeps = ((rows[..., 10].clamp(min=0) + 1) * valid).sum(1).cpu().numpy()
work = sweeps + 0.4 * roll
print("columns", TRACE_COLUMNS)
first = rows[:, 0].cpu().numpy()
order = np.argsort(-work)
print("work top 16:", [(int(b), round(float(work[b]), 1)) for b in order[:16]])
for name in TRACE_COLUMNS:
    c = first[:, TRACE_COLUMNS.index(name)]
    for sign in (1, -1):
        rk = np.argsort(np.argsort(-sign * c))            # rank 0 = largest (sign 1) / smallest
        print(f"  first-pass {name:12s} {'desc' if sign > 0 else 'asc '}: ranks of the 8 heaviest {[int(rk[b]) for b in order[:8]]}")
second = rows[:, 1].cpu().numpy()
for name in ("J_hat", "level", "alpha_index", "g_norm"):
    c = second[:, TRACE_COLUMNS.index(name)]
    rk = np.argsort(np.argsort(-c))
    print(f"  second-pass {name:12s} desc: ranks of the 8 heaviest {[int(rk[b]) for b in order[:8]]}")
x0 = w["x0"].reshape(65536, -1).cpu().numpy()
rk = np.argsort(np.argsort(-np.abs(x0).max(1)))
print("  max |x0| desc: ranks", [int(rk[b]) for b in order[:8]])
print("work histogram (>= 10, 20, 40, 80, 120):", [(int((work >= v).sum())) for v in (10, 20, 40, 80, 120)], "sum/2048 =", work.sum() / 2048)
