"""Does the stream, or the workspace allocation, change the headline kernel's time?  (The first streams a process creates
ran the cfg4 multi-stream loop 40 % slower, tools/probes/repro_cfg4_bench2.py.)  Headline LQR solve, 20 launches per
measurement, alternating the default stream and created streams, fresh and reused workspaces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.solvers.lqr import LQR
B, n, m, T = 65536, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1)
lqr = LQR(F, f, C, c)
x0 = torch.as_tensor(x0[..., None], device="cuda")
def timeit(stream, tag, ws=None):
    with torch.cuda.stream(stream):
        out = lqr.solve_device(x0, T, workspace=ws); ws = out["workspace"]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lqr.solve_device(x0, T, workspace=ws)
        e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / 20:.3f} ms (workspace at {ws.data_ptr():#x})", flush=True)
    return ws
d = torch.cuda.current_stream()
w0 = timeit(d, "default stream, fresh workspace")
timeit(d, "default stream, same workspace", w0)
timeit(d, "default stream, same workspace", w0)
s1 = torch.cuda.Stream()
timeit(s1, "created stream, same workspace", w0)
timeit(d, "default stream, same workspace", w0)
w1 = timeit(d, "default stream, second fresh workspace")
timeit(s1, "created stream, second workspace", w1)
timeit(d, "default stream, second workspace", w1)
timeit(d, "default stream, first workspace", w0)
