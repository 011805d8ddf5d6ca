"""Why does bench.py's cfg4 'sustained' loop measure 5.8 ms per batch when tools/cfg4_sustained.py measures 4.2 on the same
box?  Same loop, varying what precedes it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.navigation import Navigation
from tfmpc.solvers.ilqr import iLQR
B, T = 16384, 50
solver = iLQR(Navigation.load(problems.NAV_CONFIG))
def sustained(tag, data_on_device=True, reps=6):
    streams = [torch.cuda.Stream() for _ in range(8)]
    data = []
    for i in range(8):
        x0 = np.random.default_rng(100 + i).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
        data.append((torch.as_tensor(x0, device="cuda") if data_on_device else x0, solver.random_actions(T, B, seed=100 + i)))
    ws, outs = [None] * 8, [None] * 8
    for rep in range(1 + reps):
        if rep == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i] = solver.solve_device(data[i][0], T, u_init=data[i][1], workspace=ws[i]); ws[i] = outs[i]["workspace"]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{tag}: {dt / (8 * reps) * 1e3:.2f} ms per batch", flush=True)
    return ws
if len(sys.argv) > 1 and sys.argv[1] == "busy":
    a = torch.randn(8192, 8192, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0: b = a @ a
    torch.cuda.synchronize()
    sustained("first thing after 1 s of GEMMs")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "streams":
    old = [torch.cuda.Stream() for _ in range(8)]
    for st in old:
        with torch.cuda.stream(st): torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
    if len(sys.argv) > 2: del old
    sustained("first sustained call, after 8 throw-away streams" + (" (destroyed)" if len(sys.argv) > 2 else " (alive)"))
    sys.exit(0)
keep = sustained("first thing in the process")
keep2 = sustained("second time, the first call's workspaces still allocated (fresh memory again)")
sustained("second time")
x0 = np.random.default_rng(4).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
u0 = solver.random_actions(T, B, seed=4)
out = solver.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
for _ in range(2): out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
torch.cuda.synchronize()
sustained("after a single-batch solve with a numpy x0")
