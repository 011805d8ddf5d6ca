"""One-off fuzz of the other kernel families on random shapes (tools/probes/fuzz_costate.py covers the costate kernels):
  lqr    every LQR kernel the dispatcher picks against the fp64 C oracle (budget 10 x the fp32 restatement's error, as the tests);
  nav    Navigation iLQR: the 16-lanes-per-instance group kernel against the one-lane kernel, bit for bit;
  lq     iLQR on the LQ env (matrix-core kernels, bounded and unbounded) against the wave kernel: status equal, total cost within
         1e-3 on >= 95 % of the instances.
python tools/probes/fuzz_more.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from oracle import c_oracle
from tfmpc import _hip
from tfmpc.envs.lq import LQEnv
from tfmpc.envs.navigation import Navigation
from tfmpc.solvers.ilqr import iLQR
from tfmpc.solvers.lqr import LQR
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(77)
bad = 0
for case in range(cases):
    n, m = int(rng.integers(1, 41)), int(rng.integers(1, 25))
    T, B = int(rng.integers(0, 40)), int(rng.choice([1, 3, 31, 64, 65, 300, 2100]))
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=case)
    F = F * float(rng.choice([0.3, 0.7, 1.0])) * 2.0 / np.sqrt(n)
    ref64 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float64, nthreads=8)
    ref32 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float32, nthreads=8)
    out = LQR(F, f, C, c).solve_device(x0[..., None], T); torch.cuda.synchronize()
    ok = int(out["status"].abs().sum()) == 0
    worst = 0.0
    for key in ("states", "actions", "costs"):
        if out[key].numel() == 0: continue
        got = out[key].cpu().numpy().astype(np.float64).reshape(ref64[key].shape)
        for b in range(B):
            scale = max(np.abs(ref64[key][b]).max(), 1e-30)
            e32 = max(np.abs(ref32[key][b].astype(np.float64) - ref64[key][b]).max(), 1e-6 * scale)
            worst = max(worst, np.abs(got[b] - ref64[key][b]).max() / e32)
    ok = ok and worst <= 50
    bad += not ok
    name = _hip.load().tfmpc_lqr_kernel_name(n, m, T).decode()
    print(f"lqr case {case:3d} n={n:2d} m={m:2d} T={T:2d} B={B:4d} {name:28s} worst ratio {worst:6.2f}: {'ok' if ok else 'FAIL'}", flush=True)
env = Navigation.load(problems.NAV_CONFIG)
for case in range(cases // 2):
    B, T = int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 130, 2047, 2048, 2049, 5000])), int(rng.integers(1, 60))
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    s = iLQR(env, max_iterations=int(rng.integers(1, 12))); u0 = s.random_actions(T, B, seed=case)
    outs = {}
    for kern in ("lane1", None):
        with _hip.option("TFMPC_ILQR_KERNEL", kern):
            o = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
            outs[kern] = {k: o[k].clone() for k in ("states", "actions", "costs", "iterations", "status")}
    ok = all(torch.equal(outs["lane1"][k], outs[None][k]) for k in outs[None])
    bad += not ok
    print(f"nav case {case:3d} B={B:5d} T={T:2d}: {'ok' if ok else 'MISMATCH'}", flush=True)
for case in range(cases // 2):
    n, m = int(rng.integers(4, 33)), int(rng.integers(2, 17))
    T, B = int(rng.integers(2, 40)), int(rng.choice([1, 3, 64, 65, 300]))
    bound = None if case % 2 else float(rng.choice([0.3, 1.0, 3.0]))
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1000 + case)
    low, high = (None, None) if bound is None else (-bound, bound)
    s = iLQR(LQEnv(F * 0.2 * np.sqrt(16.0 / n), f, C, c, low=low, high=high), max_iterations=20)
    hi = 1.0 if bound is None else bound
    u0 = np.clip(0.1 * rng.normal(size=(B, T, m, 1)), -hi, hi).astype(np.float32)
    outs = {}
    for kern in ("wave", None):
        with _hip.option("TFMPC_ILQR_KERNEL", kern):
            o = s.solve_device(x0.astype(np.float32)[..., None], T, u_init=u0); torch.cuda.synchronize()
            outs[kern] = {k: o[k].clone() for k in ("costs", "iterations", "status")}
    tw, td = outs["wave"]["costs"].sum(dim=1), outs[None]["costs"].sum(dim=1)
    close = ((tw - td).abs() <= 1e-3 * tw.abs().clamp_min(1e-6)).float().mean().item()
    st = (outs["wave"]["status"] & ~_hip.ST_QP_MAXITER == outs[None]["status"] & ~_hip.ST_QP_MAXITER).float().mean().item()
    ok = close >= 0.95 and st >= 0.95
    bad += not ok
    print(f"lq  case {case:3d} n={n:2d} m={m:2d} T={T:2d} B={B:4d} bound={bound}: cost agreement {close:.2f}, status agreement {st:.2f}: {'ok' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
