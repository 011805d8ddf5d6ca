"""cfg4 (Navigation iLQR, n = m = 2, T = 50, B = 16 384): histogram of iterations per instance, what the flagged
instances are, and the batch time as a function of an iteration cap (the launch lasts as long as its slowest
instance).  Writes JSON to stdout.  Run on the GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.navigation import Navigation
from tfmpc.solvers.ilqr import iLQR

B, T = 16384, 50
env = Navigation.load(problems.NAV_CONFIG)
x0 = np.random.default_rng(4).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
res = {}
for cap in (100, 48, 32, 24, 16, 8):
    s = iLQR(env, max_iterations=cap)
    u0 = s.random_actions(T, B, seed=4)
    out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    it = (out["iterations"].cpu().numpy() + 1)
    res[f"cap_{cap}"] = {"ms": dt * 1e3, "iterations_total": int(it.sum()), "at_cap": int((it >= cap).sum()),
                         "it_per_s": float(it.sum() / dt)}
    if cap == 100:
        st = out["status"].cpu().numpy()
        hist = np.bincount(it, minlength=101)
        res["histogram_iterations"] = {str(i): int(c) for i, c in enumerate(hist) if c}
        res["quantiles"] = {q: float(np.quantile(it, float(q))) for q in ("0.5", "0.9", "0.99", "0.999")}
        bad = np.nonzero(st)[0]
        res["flagged"] = [{"instance": int(b), "status": int(st[b]), "iterations": int(it[b]), "x0": x0[b, :, 0].tolist()} for b in bad[:8]]
        res["mean_iterations"] = float(it.mean())
print(json.dumps(res, indent=1))
