"""Where a user env's iteration goes (round 6): the piecewise kernels of the reference API on hvac6 / res4 FROM PYTHON (TorchEnv.to_device_env), B = 16 384,
T = 100 -- derivatives (linearisation by dual numbers), backward pass on the materialised models, one line-search rollout -- beside the built-in env."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems, torch_envs, workloads
from tfmpc.solvers.ilqr import iLQR
B = int(os.environ.get("B", 16384))
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for name, make, cfg in (("res4", torch_envs.reservoir, problems.RES4_CONFIG), ("hvac6", torch_envs.hvac, problems.HVAC6_CONFIG)):
    w = workloads.small_env(name, B)
    res = {}
    for tag, solver in (("python", iLQR(make(dict(cfg), "cuda").to_device_env(), max_iterations=12)), ("builtin", w["solver"])):
        x0, u0, T = w["x0"], w["u0"], w["T"]
        xs, us, cs = solver.start(x0, T, u_init=u0)
        res[tag + "_start_ms"] = timed(lambda: solver.start(x0, T, u_init=u0))
        res[tag + "_derivatives_ms"] = timed(lambda: solver.derivatives(xs, us))
        models = solver.derivatives(xs, us)
        res[tag + "_backward_ms"] = timed(lambda: solver.backward(T, us, *models))
        K, k, J, dV1, dV2 = solver.backward(T, us, *models)
        res[tag + "_forward_ms"] = timed(lambda: solver.forward(xs, us, K, k))
        out = solver.solve_device(x0, T, u_init=u0, trace_rows=40)
        tr, ln = out["trace"], out["trace_len"]
        valid = torch.arange(tr.shape[1], device=tr.device)[None, :] < ln[:, None]
        res[tag + "_passes"] = float(valid.sum()); res[tag + "_rollouts"] = float((tr[..., 5] + 1)[valid & (tr[..., 8] >= 0)].sum())
        res[tag + "_solve_ms"] = timed(lambda: solver.solve_device(x0, T, u_init=u0), 2)
    print(json.dumps({name: {k: round(v, 3) for k, v in res.items()}}))
