"""Control-limited iLQR (Tassa et al. 2014, the reference's algorithm) on the headline shape: LQ env n=16, m=8, T=50
with a box on the actions, so the backward pass runs the projected-Newton box-QP at every step: the control-limited
matrix-core kernel (ilqr_lq_box_mfma.hip) against the generic wave kernel (TFMPC_ILQR_KERNEL=wave).  Run on the GPU box: python tools/bounded_lq_rate.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR

for B in [int(a) for a in sys.argv[1:]] or (1024, 8192, 65536):
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, 16, 8, seed=1)
    F *= 0.25
    for bound, kern in ((None, None), (0.5, None), (0.5, "wave")):
        if kern == "wave" and B > 8192: continue
        _hip.set_option("TFMPC_ILQR_KERNEL", kern)
        env = LQEnv(F, f, C, c) if bound is None else LQEnv(F, f, C, c, low=-bound, high=bound)
        s = iLQR(env)
        u0 = torch.zeros(B, 50, 8, 1, device="cuda")
        out = s.solve_device(x0[..., None].astype(np.float32), 50, u_init=u0); torch.cuda.synchronize()
        t = time.perf_counter()
        out = s.solve_device(x0[..., None].astype(np.float32), 50, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        its = (out["iterations"].double() + 1).sum().item()
        print(f"B={B} bounds={bound} kernel={kern or 'default'}: {dt*1e3:.2f} ms, {B/dt:.3e} solves/s, mean iterations {its/B:.1f}, {its/dt:.3e} it/s, flagged {(out['status']!=0).sum().item()}", flush=True)
        if bound is not None:
            st = out["status"].cpu().numpy(); it = out["iterations"].cpu().numpy() + 1
            names = {1: "SINGULAR", 2: "NOT_PD", 4: "NAN", 8: "QP_MAXITER", 16: "MAX_ATTEMPTS"}
            print("   status bits:", {v: int(((st & k) != 0).sum()) for k, v in names.items()},
                  "iterations p50/p90/max:", int(np.median(it)), int(np.quantile(it, 0.9)), int(it.max()))
