"""The heaviest instances of bench.py's stable control-limited workload: their decision traces (what the straggler that sets the launch time does)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
from tfmpc.solvers.ilqr import trace_records
w = workloads.control_limited_stable(65536)
out = workloads.solver_of(w).solve_device(w["x0"], w["T"], u_init=w["u0"], trace_rows=170)
torch.cuda.synchronize()
ln = out["trace_len"].cpu().numpy(); it = out["iterations"].cpu().numpy(); st = out["status"].cpu().numpy()
rows = torch.nan_to_num(out["trace"], nan=-1.0)
valid = torch.arange(rows.shape[1], device=rows.device)[None, :] < out["trace_len"][:, None]
searched = valid & (rows[..., 8] >= 0)
roll = ((rows[..., 5].clamp(min=0) + 1) * searched).sum(1).cpu().numpy()
sweeps = ((rows[..., 10].clamp(min=0) + 1) * valid).sum(1).cpu().numpy()
work = sweeps + 0.4 * roll
order = np.argsort(-work)
print("work (sweep equivalents) top 12:", [(int(b), float(work[b]), int(it[b]) + 1, int(ln[b]), int(st[b])) for b in order[:12]])
print("work quantiles p50 p99 p99.9 p99.99:", np.quantile(work, [0.5, 0.99, 0.999, 0.9999]).tolist(), "sum/2048:", work.sum() / 2048)
for b in order[:3]:
    recs = trace_records(out["trace"][[int(b)]], out["trace_len"][[int(b)]])[0]
    print(f"--- instance {int(b)}: {len(recs)} passes, iterations {int(it[b]) + 1}, status {int(st[b])}")
    for p, r in enumerate(recs[:6] + recs[-6:]):
        print("   ", {k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()})
    import collections
    print("    levels:", collections.Counter(r["level"] for r in recs), "alpha_index:", collections.Counter(r["alpha_index"] for r in recs),
          "accepted:", collections.Counter(r["accepted"] for r in recs))
    # the pass-by-pass pattern (A = accepted, r = rejected with all step sizes tried) and the regularisation level each sweep ended on: runs of
    # rejections are what a speculative sweep of the NEXT mu could overlap
    print("    pattern:", "".join("A" if r["accepted"] == 1 else "r" for r in recs))
    print("    levels :", "".join(str(min(int(r["level"]), 9)) for r in recs))
