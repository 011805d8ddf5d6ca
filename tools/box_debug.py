"""Debug helper: bounded LQ solves, matrix-core kernel vs wave kernel vs oracle on the instances where they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from oracle import envs_ref, ilqr_ref
from tfmpc import _hip
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
n, m, T, bound, B = 16, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 30, float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, 96
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=10 * n + m); F = F * 0.25; x0 = x0.astype(np.float32)
solver = iLQR(LQEnv(F, f, C, c, low=-bound, high=bound))
u0 = np.clip(0.1 * np.random.default_rng(1).normal(size=(B, T, m, 1)), -bound, bound).astype(np.float32)
out = {}
for kern in (None, "wave"):
    _hip.set_option("TFMPC_ILQR_KERNEL", kern)
    out[kern] = solver.solve_device(x0[..., None], T, u_init=u0); torch.cuda.synchronize()
mf, wv = out[None], out["wave"]
tm, tw = mf["costs"].sum(1).cpu().numpy(), wv["costs"].sum(1).cpu().numpy()
rel = np.abs(tm - tw) / np.abs(tw)
print('mfma worse by >1%:', int(((tm - tw) / np.abs(tw) > 1e-2).sum()), ' better by >1%:', int(((tw - tm) / np.abs(tw) > 1e-2).sum()))
for b in np.argsort(-rel)[:7]:
    o = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b], low=-bound, high=bound))
    x, u, cs, it = o.solve(x0[b], T, u_init=u0[b])
    o32 = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b], low=-bound, high=bound, dtype=np.float32), dtype=np.float32)
    try:
        x32, u32, c32, it32 = o32.solve(x0[b], T, u_init=u0[b]); c32s = float(c32.sum())
    except Exception as e:
        c32s, it32 = repr(e)[:40], -1
    print(f"b={b}: mfma cost {tm[b]:.4f} it {int(mf['iterations'][b])} st {int(mf['status'][b])} | wave cost {tw[b]:.4f} it {int(wv['iterations'][b])} st {int(wv['status'][b])}"
          f" | fp64 {cs.sum():.4f} it {it} mus {[round(r['mu'],6) for r in o.trace][-6:]} | fp32 {c32s} it {it32}")
