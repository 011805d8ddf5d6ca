"""Debug probe: one control-limited instance -- gains of the first backward pass of the matrix-core box kernel against the fp64 / fp32
restatement, step by step (which time step's box-QP ends on another free set?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import workloads
from oracle import envs_ref, ilqr_ref
from tfmpc import _hip
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR, trace_records

b = int(sys.argv[1]) if len(sys.argv) > 1 else 68
w = workloads.control_limited(65536)
n, m, T = 16, 8, w["T"]
env = LQEnv(w["F"][b], w["f"][b], w["C"][b], w["c"][b], low=w["low"], high=w["high"])
x0, u0 = w["x0"][b:b + 1], w["u0"][b:b + 1]
gains = {}
for kern in (None, "wave"):
    with _hip.option("TFMPC_ILQR_KERNEL", kern):
        solver = iLQR(env, max_iterations=1)
        out = solver.solve_device(x0, T, u_init=u0, trace_rows=8)
        torch.cuda.synchronize()
    ws = out["workspace"].view(torch.float32) if out["workspace"].dtype != torch.float32 else out["workspace"]
    K = ws[:T * m * n].reshape(T, m, n).cpu().numpy().astype(np.float64)
    k = ws[T * m * n:T * m * n + T * m].reshape(T, m).cpu().numpy().astype(np.float64)
    gains[kern] = (K, k)
    print(kern, [(r["alpha_index"], r["J_hat"], r["J"], r["residual"], r["g_norm"]) for r in trace_records(out["trace"], out["trace_len"])[0]])
ref = {}
for dt in (np.float64, np.float32):
    o = ilqr_ref.ILQRRef(envs_ref.LQEnv(w["F"][b], w["f"][b], w["C"][b], w["c"][b], low=w["low"], high=w["high"], dtype=dt), dtype=dt)
    xs, us, cs = o.start(x0[0].cpu().numpy(), T, u_init=u0[0].cpu().numpy())
    models = o.derivatives(xs, us)
    K, k, J, dV1, dV2 = o.backward(T, us, *models, mu=0.0)
    x, u, c, Jn, res = o.forward(xs, us, K, k, dt(1.0))
    ref[dt] = (K.astype(np.float64), k[..., 0].astype(np.float64))
    print(dt.__name__, "J_hat", float(J), "J", float(Jn), "residual", float(res))
K64, k64 = ref[np.float64]
for name, (K, k) in (("box-mfma", gains[None]), ("wave", gains["wave"]), ("ref32", ref[np.float32])):
    dk = np.abs(k - k64).max(axis=1)
    dK = np.abs(K - K64).max(axis=(1, 2))
    zero_rows = [(t, int(((np.abs(K[t]).max(axis=1) == 0) != (np.abs(K64[t]).max(axis=1) == 0)).sum())) for t in range(T)]
    bad = [(t, round(float(dk[t]), 5), round(float(dK[t]), 5), z) for (t, z) in zero_rows if z or dk[t] > 1e-3 or dK[t] > 1e-3]
    print(name, "max |dk|", float(dk.max()), "max |dK|", float(dK.max()), "steps that differ (t, |dk|, |dK|, clamped rows that differ):", bad[:20])

# accepted candidate of pass 0 (max_iterations = 1 -> it is the output): trajectories and costs per kernel against the fp64 restatement
alphas = np.geomspace(1.0, 1e-3, 11)
o = ilqr_ref.ILQRRef(envs_ref.LQEnv(w["F"][b], w["f"][b], w["C"][b], w["c"][b], low=w["low"], high=w["high"]))
xs, us, cs = o.start(x0[0].cpu().numpy(), T, u_init=u0[0].cpu().numpy())
K, k, J, dV1, dV2 = o.backward(T, us, *o.derivatives(xs, us), mu=0.0)
F64, f64, C64, c64 = (np.asarray(w[key][b], dtype=np.float32).astype(np.float64) for key in ("F", "f", "C", "c"))
for ai in (2,):
    x, u, c, Jn, res = o.forward(xs, us, K, k, alphas[ai])
    for kern in (None, "wave"):
        with _hip.option("TFMPC_ILQR_KERNEL", kern):
            out = iLQR(env, max_iterations=1).solve_device(x0, T, u_init=u0)
            torch.cuda.synchronize()
        X = out["states"][0, :, :, 0].cpu().numpy().astype(np.float64)
        U = out["actions"][0, :, :, 0].cpu().numpy().astype(np.float64)
        Cst = out["costs"][0].cpu().numpy().astype(np.float64)
        z = np.concatenate([X[:-1], U], axis=1)
        recomputed = 0.5 * np.einsum("ti,ij,tj->t", z, C64, z) + z @ c64
        print(kern, "alpha", alphas[ai], "max |x - x64|", np.abs(X - x[..., 0]).max(), "max |u - u64|", np.abs(U - u[..., 0]).max(),
              "sum costs", Cst.sum(), "fp64 J", float(Jn), "| device stage costs vs fp64 re-evaluation on the device trajectory: max abs",
              np.abs(Cst[:-1] - recomputed).max(), "sum", Cst[:-1].sum() - recomputed.sum(),
              "| per-step |x - x64| growth", np.round(np.abs(X - x[..., 0]).max(axis=1)[::7], 4))
