"""Host time of one iLQR.solve_device call (no synchronisation: what the Python side spends before the launch is queued), by cProfile, for a small-env solve
(res4, B = 16 384) where the kernel is 1.4 ms: how much of the wall-clock gap between device time and per-call time is the host's."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
B, T = 16384, 100
env = Reservoir.load(dict(problems.RES4_CONFIG))
x = torch.as_tensor(np.tile(np.array(problems.RES4_X0, dtype=np.float32)[None], (B, 1, 1)), device="cuda")
s = iLQR(env, max_iterations=12); u0 = torch.as_tensor(s.random_actions(T, B, seed=1), device="cuda")
out = s.solve_device(x, T, u_init=u0); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): out = s.solve_device(x, T, u_init=u0, workspace=out["workspace"])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host time per call (queued, not waited for): {(t1 - t0) / 200 * 1e6:.1f} us; wall per call incl. device: {(t2 - t0) / 200 * 1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): out = s.solve_device(x, T, u_init=u0, workspace=out["workspace"])
pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(14); print(st.getvalue()[:2600])
