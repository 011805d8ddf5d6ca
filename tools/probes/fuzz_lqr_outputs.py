"""One-off: all eight outputs (states, actions, costs, K, k, V, v, const) of every LQR kernel the dispatcher picks, on random
shapes, against the fp64 C oracle (ratio to the fp32 restatement's own error; the tests allow 50 at the tail).
python tools/probes/fuzz_lqr_outputs.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from oracle import c_oracle
from tfmpc import _hip
from tfmpc.solvers.lqr import LQR
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(123)
bad = 0
for case in range(cases):
    n, m = int(rng.integers(1, 41)), int(rng.integers(1, 25))
    T, B = int(rng.integers(1, 30)), int(rng.choice([1, 3, 33, 64, 65, 300]))
    gen = problems.make_lqr_batch_spd if case % 3 == 0 else problems.make_lqr_batch_fast
    F, f, C, c, x0 = gen(B, n, m, seed=case)
    F = F * float(rng.choice([0.3, 0.7])) * 2.0 / np.sqrt(n)
    ref64 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float64, nthreads=8, want_policy=True, want_value=True)
    ref32 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float32, nthreads=8, want_policy=True, want_value=True)
    out = LQR(F, f, C, c).solve_device(x0[..., None], T, want_policy=True, want_value=True); torch.cuda.synchronize()
    flagged = int((out["status"] != 0).sum())
    worst, where = 0.0, ""
    for key in ("states", "actions", "costs", "K", "k", "V", "v", "const"):
        got = out[key].cpu().numpy().astype(np.float64).reshape(ref64[key].shape)
        for b in range(B):
            if out["status"][b] != 0: continue
            scale = max(np.abs(ref64[key][b]).max(), 1e-30)
            e32 = max(np.abs(ref32[key][b].astype(np.float64) - ref64[key][b]).max(), 1e-6 * scale)
            r = np.abs(got[b] - ref64[key][b]).max() / e32
            if r > worst: worst, where = r, key
    ok = worst <= 50 and flagged <= B // 10
    bad += not ok
    name = _hip.load().tfmpc_lqr_kernel_name(n, m, T).decode()
    print(f"case {case:3d} n={n:2d} m={m:2d} T={T:2d} B={B:3d} {'spd ' if case % 3 == 0 else 'fast'} {name[:24]:24s} worst ratio {worst:6.2f} ({where}), flagged {flagged}: {'ok' if ok else 'FAIL'}", flush=True)
print("failures:", bad); sys.exit(1 if bad else 0)
