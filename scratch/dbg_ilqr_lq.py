import os, sys
sys.path.insert(0, '/root/repo/tf-mpc_amd'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np, torch, problems
from oracle import envs_ref, ilqr_ref
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
B, n, m, T = 80, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=100 * n + m)
F = F * 0.25; x0 = x0.astype(np.float32)
solver = iLQR(LQEnv(F, f, C, c))
u0 = (0.1 * np.random.default_rng(1).normal(size=(B, T, m, 1))).astype(np.float32)
out = {}
for kern in (None, "wave"):
    if kern: os.environ["TFMPC_ILQR_KERNEL"] = kern
    else: os.environ.pop("TFMPC_ILQR_KERNEL", None)
    out[kern] = solver.solve_device(x0[..., None], T, u_init=u0); torch.cuda.synchronize()
mf, wv = out[None], out["wave"]
err = (mf["states"] - wv["states"]).abs().amax(dim=(1, 2, 3)).cpu().numpy()
print("per-instance max err quantiles", np.quantile(err, [0.5, 0.9, 0.99, 1.0]), "argmax", err.argmax())
b = int(err.argmax())
o = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b]))
x, u, cs, it = o.solve(x0[b], T, u_init=u0[b])
print("it oracle", it, "mfma", int(mf["iterations"][b]), "wave", int(wv["iterations"][b]))
for name, r in (("mfma", mf), ("wave", wv)):
    e = np.abs(r["states"][b, ..., 0].cpu().numpy() - x)
    print(name, "vs fp64: max err", e.max(), "at t", np.unravel_index(e.argmax(), e.shape), "scale", np.abs(x).max(), "cost", float(r["costs"][b].sum()), cs.sum())
o32 = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b], dtype=np.float32), dtype=np.float32)
x32, u32, c32, it32 = o32.solve(x0[b], T, u_init=u0[b])
print("fp32 oracle vs fp64: max err", np.abs(x32 - x).max(), "it", it32)
