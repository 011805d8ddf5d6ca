"""Generic-env path (TorchEnv, host-driven loop): one batched solve with the rollout / derivative blocks replayed as hipGraphs
(`iLQR(env, graphs=True)`, the default) and eagerly; same results required.  python tools/probes/torchenv_graphs.py [B] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.torchenv import TorchEnv
from tfmpc.solvers.ilqr import iLQR
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = problems.NAV_CONFIG
goal = torch.tensor(np.array(cfg["goal"], dtype=np.float32).reshape(-1), device="cuda")
centers = torch.tensor(np.array(cfg["deceleration"]["center"], dtype=np.float32).reshape(-1, 2), device="cuda")
decay = torch.tensor(np.array(cfg["deceleration"]["decay"], dtype=np.float32).reshape(-1), device="cuda")
def transition(x, u):
    r = torch.linalg.norm(x[None, :] - centers, dim=1)
    lam = torch.prod(2.0 / (1.0 + torch.exp(-decay * r)) - 1.0)
    return x + lam * u
def cost(x, u):
    return torch.sum((x - goal) ** 2)
env = TorchEnv(transition, cost, lambda x: torch.sum((x - goal) ** 2), 2, 2, auto_compile=False,
               low=np.array(cfg["low"], dtype=np.float32).reshape(-1, 1), high=np.array(cfg["high"], dtype=np.float32).reshape(-1, 1))
rng = np.random.default_rng(4)
x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
outs = {}
for graphs in (False, True):
    solver = iLQR(env, max_iterations=10, graphs=graphs)
    u0 = solver.random_actions(T, B, seed=4)
    solver.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out = solver.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    outs[graphs] = out
    its = float((out["iterations"].double() + 1).sum())
    eager = [k for k, g in solver._graphed.items() if g.eager]
    print(f"graphs={graphs!s:5}: {min(ts):7.1f} ms per batch of {B} (T = {T}), {its / min(ts) * 1e3:9.0f} iterations/s, blocks that fell back to eager: {eager}")
same = all(torch.equal(outs[False][k], outs[True][k]) for k in ("states", "actions", "costs", "iterations", "status"))
print("identical results:", same, "| max |dx|", float((outs[False]["states"] - outs[True]["states"]).abs().max()))
