import sys, time
sys.path.insert(0, '/root/repo/tf-mpc_amd'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np, torch, problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
n, T, B = 32, 100, 8192
rng = np.random.default_rng(4)
names = ["bwd load+wait", "bwd cost", "bwd adjoint Qx/Qu", "bwd reductions+copy", "fwd load+wait", "fwd u/clip/max", "fwd cost", "fwd transition", "fwd store"]
for kind in ("hvac", "reservoir"):
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=4); u0 = s.random_actions(T, B, seed=5)
    out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    prof = out["workspace"][:32].view(torch.int64)[:9].cpu().numpy()
    tot = prof.sum()
    print(kind, "cycles of wave 0 over 4 iterations:", int(tot))
    for nm, v in zip(names, prof):
        if nm: print(f"   {nm:18s} {v/tot*100:5.1f} %   {v/ (4*100):9.0f} cycles per step-iteration")
