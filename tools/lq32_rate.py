"""BASELINE configs[4] at its LITERAL dims as iLQR on the generalised LQ env: n = 32, m = 16, T = 100.
Times iLQR.solve on LQEnv at a few batch sizes.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
n, m, T = 32, 16, 100
for B in [int(a) for a in sys.argv[1:]] or [1024, 8192]:
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1)
    s = iLQR(LQEnv(F * (0.9 / np.sqrt(n)), f, C, c))
    x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda"); u0 = torch.zeros(B, T, m, 1, device="cuda")
    out = s.solve_device(x0d, T, u_init=u0); torch.cuda.synchronize()
    t = time.perf_counter(); out = s.solve_device(x0d, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    it = float((out["iterations"].double() + 1).sum())
    print(f"iLQR LQEnv n={n} m={m} T={T} B={B}: {dt*1e3:.1f} ms, {it/B:.2f} iterations/instance, {it/dt/1e6:.3f} M it/s, flagged {int((out['status']!=0).sum())}", flush=True)
