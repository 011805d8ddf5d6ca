"""cfg4 (Navigation iLQR, n = m = 2, T = 50, B = 16 384 per batch): single-batch time, and SUSTAINED throughput with
several batches in flight on their own streams (a single launch lasts as long as its slowest instance; other
batches fill the chip meanwhile).  JSON to stdout.  Run on the GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.navigation import Navigation
from tfmpc.solvers.ilqr import iLQR

B, T = 16384, 50
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3          # timed repetitions per stream
solver = iLQR(Navigation.load(problems.NAV_CONFIG))
res = {"single_batch": {}, "sustained": {}}


def batch(seed):
    x0 = torch.as_tensor(np.random.default_rng(seed).uniform(0, 10, size=(B, 2, 1)).astype(np.float32), device="cuda")
    return x0, solver.random_actions(T, B, seed=seed)


x0, u0 = batch(4)
for park in ("-",):
    out = solver.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    its = float((out["iterations"].double() + 1).sum())
    res["single_batch"] = {"ms": dt * 1e3, "it_per_s": its / dt}
for n_streams in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    data = [batch(100 + i) for i in range(n_streams)]
    ws = [None] * n_streams
    outs = [None] * n_streams
    for rep in range(1 + REPS):               # rep 0 = warm-up (allocates the workspaces)
        if rep == 1:
            torch.cuda.synchronize(); t = time.perf_counter()
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs[i] = solver.solve_device(data[i][0], T, u_init=data[i][1], workspace=ws[i])
                ws[i] = outs[i]["workspace"]
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    its = sum(float((o["iterations"].double() + 1).sum()) for o in outs) * REPS
    res["sustained"][f"{n_streams}_batches_in_flight"] = {"ms_per_batch": dt / (REPS * n_streams) * 1e3, "it_per_s": its / dt}
print(json.dumps(res, indent=1))
