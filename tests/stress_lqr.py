"""Randomised parity sweep of the LQR kernels against the C oracle (fp64 = truth, fp32 = the
noise floor of an fp32 implementation in the reference's operation order): shapes n <= 16, m <= 8 (matrix-core
kernel; `python tests/stress_lqr.py [cases] large` draws n <= 40, m <= 24 instead: block and wave kernels),
horizons 1..60, well- and ill-conditioned costs (make_spd_matrix-like spectra down to 0.02), unstable F.
Reports, per case, the device error relative to the fp32 restatement's error.  Run on the GPU box:
python tests/stress_lqr.py [cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch
from oracle import c_oracle
from tfmpc.solvers.lqr import LQR

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
large = len(sys.argv) > 2 and sys.argv[2] == "large"
rng = np.random.default_rng(2024)
c_oracle.build()
worst = []
t_start = time.time()
for case in range(cases):
    n = int(rng.integers(3, 17)); m = int(rng.integers(1, 9))
    if large: n = int(rng.integers(6, 41)); m = int(rng.integers(1, 25))
    if n + m <= 6: n = 7 - m
    T = int(rng.integers(1, 61)); B = int(rng.integers(40, 400)) if not large else int(rng.choice([3, 40, 300, 2500]))
    d = n + m
    rho = rng.choice([0.5, 1.0, 2.0, 4.0])                      # spectral scale of F
    lam_min = rng.choice([1.0, 0.2, 0.04, 0.02])                 # smallest eigenvalue of C
    F = rng.normal(size=(B, n, d)) * rho / np.sqrt(n)
    f = rng.normal(size=(B, n)); c = rng.normal(size=(B, d))
    Q, _ = np.linalg.qr(rng.normal(size=(B, d, d)))
    ev = lam_min + rng.uniform(size=(B, d)) * rng.choice([1.0, 10.0, d])
    C = np.einsum("bik,bk,bjk->bij", Q, ev, Q); C = 0.5 * (C + np.swapaxes(C, 1, 2))
    x0 = rng.normal(size=(B, n))
    ref64 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float64, nthreads=8, want_policy=True)
    ref32 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float32, nthreads=8, want_policy=True)
    out = LQR(F, f, C, c).solve_device(x0[..., None], T, want_policy=True)
    torch.cuda.synchronize()
    flagged = int((out["status"] != 0).sum())
    line = [f"case {case:3d} n={n:2d} m={m} T={T:2d} B={B:3d} rho={rho} lam_min={lam_min}: flagged {flagged}"]
    for key in ("states", "actions", "costs", "K"):
        got = out[key].cpu().numpy().reshape(ref64[key].shape).astype(np.float64)
        scale = np.abs(ref64[key]).reshape(B, -1).max(axis=1) + 1e-30
        e_dev = np.abs(got - ref64[key]).reshape(B, -1).max(axis=1) / scale
        e_32 = np.abs(ref32[key].astype(np.float64) - ref64[key]).reshape(B, -1).max(axis=1) / scale
        finite = np.isfinite(e_32) & np.isfinite(scale) & (scale < 1e30)
        ratio = np.median(e_dev[finite]) / max(np.median(e_32[finite]), 1e-9) if finite.any() else float("nan")
        line.append(f"{key}: dev med {np.median(e_dev[finite]):.1e} p99 {np.quantile(e_dev[finite], 0.99):.1e} | fp32 med {np.median(e_32[finite]):.1e} | ratio {ratio:.2f}")
        worst.append((ratio, case, key))
    print("  ".join(line), flush=True)
worst = [w for w in worst if np.isfinite(w[0])]
worst.sort(reverse=True)
print("largest device/fp32 median-error ratios:", [(round(r, 2), c_, k) for r, c_, k in worst[:6]])
print(f"{cases} cases in {time.time() - t_start:.0f} s")
