import os, sys
ROOT = "/root/repo"
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR, TRACE_COLUMNS
n, m, T, B = 16, 8, 50, 96
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=7 * n + m)
F = F * 0.25; x0 = x0.astype(np.float32)
u0 = (0.1 * np.random.default_rng(2).normal(size=(B, T, m, 1))).astype(np.float32)
solver = iLQR(LQEnv(F, f, C, c), atol=1e-12, max_iterations=6)
out = {}
for mode in (None, "0"):
    with _hip.option("TFMPC_ILQR_LQ_REUSE", mode):
        out[mode] = solver.solve_device(x0[..., None], T, u_init=u0, trace_rows=8)
    torch.cuda.synchronize()
a, b = out[None]["trace"][:, 0], out["0"]["trace"][:, 0]
d = (a - b).abs().nan_to_num(0)
print(TRACE_COLUMNS)
print("max diff per col", d.amax(0).tolist())
bad = d.amax(1).nonzero().flatten().tolist()
print("rows differing", len(bad), bad[:10])
for r in bad[:3]:
    print(a[r].tolist()); print(b[r].tolist())
print(out[None]["trace"][0,:7].tolist())
print(out["0"]["trace"][0,:7].tolist())
