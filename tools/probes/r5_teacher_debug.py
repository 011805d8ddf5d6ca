"""Where inside a backward pass does the control-limited kernel part from the restatement?  For (instance, pass) pairs of the
`control_limited` workload (sub-batch order of tests/test_ilqr_teacher_forced_gpu.py): the device's gains K_t, k_t of that pass (left
in the workspace by a launch with max_iterations = iteration + 1 whose last pass it is) against the fp32 / fp64 restatement's from the
device's own nominal trajectory, step by step from t = T - 1 down; at the first step that differs, the box-QP's data.
    python tools/probes/r5_teacher_debug.py 1:1 38:4 65:4 72:0 8:0"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import workloads, trace_oracle, teacher_forced as tf
from oracle import ilqr_ref, boxqp_ref
from tfmpc import _hip
from tfmpc.solvers.ilqr import trace_records

np.set_printoptions(precision=6, linewidth=200, suppress=False)
pairs = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(1, 1)]
w = workloads.control_limited(65536)
rows = 170
full = workloads.solver_of(w).solve_device(w["x0"], w["T"], u_init=w["u0"], trace_rows=rows)
torch.cuda.synchronize()
st, it = full["status"].cpu().numpy(), full["iterations"].cpu().numpy()
retried = np.flatnonzero((st & _hip.ST_NOT_PD) != 0)
capped = np.flatnonzero((st & _hip.ST_MAX_ATTEMPTS) != 0)
family = np.flatnonzero((it == 99) & ((st & _hip.ST_MAX_ATTEMPTS) == 0) & ((st & _hip.ST_NOT_PD) != 0))
light = retried[np.argsort(it[retried], kind="stable")][:16]
pick = []
for g in (np.arange(80), light, family[:16], capped[:16]):
    pick += [int(b) for b in g if int(b) not in pick]
T, n, m = w["T"], 16, 8
for i, p in pairs:
    b = pick[i]
    one = dict(w, F=w["F"][[b]], f=w["f"][[b]], C=w["C"][[b]], c=w["c"][[b]], x0=w["x0"][[b]].contiguous(), u0=w["u0"][[b]].contiguous())
    tr = trace_records(full["trace"][[b]], full["trace_len"][[b]])[0]
    d = tr[p]
    k_it = d["iteration"]
    print(f"=== sub-batch instance {i} (batch index {b}), pass {p}: {d}")
    if k_it == 0:
        nom = workloads.solver_of(one, max_iterations=1, atol=1e9).solve_device(one["x0"], T, u_init=one["u0"])
    else:
        nom = workloads.solver_of(one, max_iterations=k_it).solve_device(one["x0"], T, u_init=one["u0"])
    out = workloads.solver_of(one, max_iterations=k_it + 1).solve_device(one["x0"], T, u_init=one["u0"], trace_rows=rows)
    torch.cuda.synchronize()
    last = trace_records(out["trace"], out["trace_len"])[0]
    if len(last) != p + 1:
        print(f"  pass {p} is not the last pass of the max_iterations={k_it + 1} run ({len(last)} passes): gains in the workspace are another pass's; skipped")
        continue
    ws = out["workspace"]
    K_dev = ws[:T * m * n].reshape(T, m, n).cpu().numpy().astype(np.float64)
    k_dev = ws[T * m * n:T * m * n + T * m].reshape(T, m).cpu().numpy().astype(np.float64)
    x_hat = nom["states"][0].cpu().numpy().astype(np.float64)
    u_hat = nom["actions"][0].cpu().numpy().astype(np.float64)
    cfg = workloads.instance_cfg(w, b)
    res = {}
    for name, dtype in (("fp32", np.float32), ("fp64", np.float64)):
        o = ilqr_ref.ILQRRef(trace_oracle.make_env("lq", cfg, dtype), dtype=dtype)
        models = o.derivatives(x_hat.astype(dtype), u_hat.astype(dtype))
        mu_l = tf._bumped(o, d["mu"], d["delta"], d["level"])
        K, k, J, dV1, dV2 = o.backward(T, u_hat.astype(dtype), *models, dtype(mu_l))
        res[name] = (K.astype(np.float64), k[..., 0].astype(np.float64))
    K64, k64 = res["fp64"]
    K32, k32 = res["fp32"]
    low, high = -0.5 - u_hat[..., 0], 0.5 - u_hat[..., 0]
    for t in range(T - 1, -1, -1):
        ek, ek32 = np.abs(k_dev[t] - k64[t]).max(), np.abs(k32[t] - k64[t]).max()
        eK, eK32 = np.abs(K_dev[t] - K64[t]).max(), np.abs(K32[t] - K64[t]).max()
        flag = ek > max(20 * ek32, 1e-3) or eK > max(20 * eK32, 1e-2 * max(np.abs(K64[t]).max(), 1e-3))
        if flag or t >= T - 2:
            print(f"  t={t}: |k_dev-k64| {ek:.3e} (fp32 restatement {ek32:.3e})   |K_dev-K64| {eK:.3e} (fp32 {eK32:.3e}) scale K {np.abs(K64[t]).max():.3e}")
        if flag:
            print("   k_dev ", k_dev[t]); print("   k_64  ", k64[t]); print("   k_32  ", k32[t])
            print("   low   ", low[t]); print("   high  ", high[t])
            cl_dev = (np.abs(k_dev[t] - low[t]) < 1e-6) | (np.abs(k_dev[t] - high[t]) < 1e-6)
            cl_64 = (np.abs(k64[t] - low[t]) < 1e-6) | (np.abs(k64[t] - high[t]) < 1e-6)
            print("   at bound dev", cl_dev.astype(int), " fp64", cl_64.astype(int))
            print("   rows of K that are zero: dev", (np.abs(K_dev[t]).max(1) == 0).astype(int), " fp64", (np.abs(K64[t]).max(1) == 0).astype(int))
            break
