"""LQR.solve at the headline shape (n = 16, m = 8, T = 50, B = 65 536) WITH the policy and value-function outputs
(lqr.py:107-129: K, k, V, v, const = 81.8 KB per solve on top of the 5 KB trajectory), fp32 containers or -- argument
"bf16" -- 16-bit containers: one warm-up + 3 launches, for rocprofv3 (WRITE_SIZE shows the bytes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.solvers.lqr import LQR
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
B, n, m, T = 65536, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=1234)
lqr = LQR(F, f, C, c)
x0d = lqr._prep_x0(x0)
out = lqr.solve_device(x0d, T, want_policy=True, want_value=True, storage_bf16=bf16); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): out = lqr.solve_device(x0d, T, want_policy=True, want_value=True, storage_bf16=bf16, workspace=out["workspace"])
torch.cuda.synchronize()
nbytes = sum(out[k].numel() * out[k].element_size() for k in ("K", "k", "V", "v", "const"))
print(f"{'bf16' if bf16 else 'fp32'} outputs: {(time.perf_counter() - t) / 3 * 1e3:.3f} ms per launch, policy + value arrays {nbytes / 1e9:.3f} GB, flagged {int((out['status'] != 0).sum())}")
