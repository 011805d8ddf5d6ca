"""cfg4 (Navigation, B = 16 384): which step size does a pass adopt?  From the decision trace: histogram of `alpha_index` over all passes, and over the
passes of the slowest instances (the launch lasts as long as they do)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
from tfmpc.solvers.ilqr import TRACE_COLUMNS
w = workloads.cfg4(1)
s, x0, u0 = w["solver"], w["x0"], w["u0"]
out = s.solve_device(x0, w["T"], u_init=u0, trace_rows=128)
torch.cuda.synchronize()
tr = out["trace"].cpu().numpy(); ln = out["trace_len"].cpu().numpy()
ia, iacc = TRACE_COLUMNS.index("alpha_index"), TRACE_COLUMNS.index("accepted")
valid = np.arange(tr.shape[1])[None, :] < ln[:, None]
a = tr[..., ia]; acc = tr[..., iacc]
searched = valid & (a >= 0)
print("kernel", s.last_kernel, "passes", int(valid.sum()), "with a line search", int(searched.sum()))
print("adopted index histogram (all):", np.bincount(a[searched].astype(int), minlength=11).tolist(), "accepted share", float((acc[searched] == 1).mean()))
slow = np.argsort(-ln)[:32]
m = searched[slow]
print("passes of the 32 slowest instances:", ln[slow].tolist())
print("adopted index histogram (32 slowest):", np.bincount(a[slow][m].astype(int), minlength=11).tolist(), "accepted share", float((acc[slow][m] == 1).mean()))
