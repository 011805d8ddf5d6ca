"""Control-limited kernel A/B: results of two builds on the same 65 536 problems (TFMPC_LIB selects the build per process):
python tools/probes/box_ab.py save|compare <file.npz>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
B = 65536
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, 16, 8, seed=4321)
s = iLQR(LQEnv(0.25 * F, f, C, c, low=-0.5, high=0.5))
u0 = torch.zeros(B, 50, 8, 1, device="cuda")
x0 = x0[..., None].astype(np.float32)
out = s.solve_device(x0, 50, u_init=u0); torch.cuda.synchronize()
t = time.perf_counter(); out = s.solve_device(x0, 50, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
print(f"{(time.perf_counter() - t) * 1e3:.1f} ms per {B} solves")
res = {k: out[k].cpu().numpy() for k in ("states", "actions", "costs", "iterations", "status")}
st = res["status"]
print("capped", int(((st & 16) != 0).sum()), "with retries", int(((st & 2) != 0).sum()), "max iterations", int(res["iterations"].max()) + 1)
if sys.argv[1] == "save":
    np.savez(sys.argv[2], **res)
else:
    ref = np.load(sys.argv[2])
    uncapped = ((ref["status"] & 16) == 0) & ((st & 16) == 0)
    same = np.array([np.array_equal(res[k][uncapped], ref[k][uncapped]) for k in ("states", "actions", "costs", "iterations")])
    per = (np.abs(res["states"] - ref["states"]).reshape(B, -1).max(axis=1) == 0)
    diff = uncapped & ~per
    tc, tr = res["costs"].reshape(B, -1).sum(axis=1), ref["costs"].reshape(B, -1).sum(axis=1)
    rel = (tc[diff] - tr[diff]) / np.abs(tr[diff])
    if diff.any():
        print("un-capped instances whose trajectory differs:", int(diff.sum()), "| total cost, this build relative to the other: median",
              float(np.median(rel)), "p10", float(np.quantile(rel, 0.1)), "p90", float(np.quantile(rel, 0.9)), "worse by > 1 %:", int((rel > 0.01).sum()),
              "better by > 1 %:", int((rel < -0.01).sum()), "| iterations", float(res["iterations"][diff].mean()), "vs", float(ref["iterations"][diff].mean()))
    print("un-capped instances:", int(uncapped.sum()), "all outputs bit-identical:", bool(same.all()),
          "| instances with identical states:", int(per.sum()), "of", B,
          "| status differs on", int((st != ref["status"]).sum()))
