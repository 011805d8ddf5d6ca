"""How does the accepted step-size index of one instance move from pass to pass (res4 / hvac6, B = 16 384, T = 100, 12 iterations)?  Histogram of
chosen - (index accepted last): what a guess-based candidate store hits."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
from tfmpc.solvers.ilqr import TRACE_COLUMNS
for name in ("res4", "hvac6"):
    w = workloads.small_env(name)
    out = w["solver"].solve_device(w["x0"], w["T"], u_init=w["u0"], trace_rows=16)
    torch.cuda.synchronize()
    tr = out["trace"].cpu().numpy(); ln = out["trace_len"].cpu().numpy()
    ia, iacc = TRACE_COLUMNS.index("alpha_index"), TRACE_COLUMNS.index("accepted")
    moves = collections.Counter()
    last = np.zeros(tr.shape[0], dtype=int)
    for p in range(int(ln.max())):
        ok = (p < ln) & (tr[:, p, ia] >= 0)
        d = tr[ok, p, ia].astype(int) - last[ok]
        for v, c in zip(*np.unique(d, return_counts=True)): moves[int(v)] += int(c)
        acc = ok & (tr[:, p, iacc] == 1)
        last[acc] = tr[acc, p, ia].astype(int)
    tot = sum(moves.values())
    print(name, "chosen - guess:", {k: round(v / tot, 3) for k, v in sorted(moves.items())})
