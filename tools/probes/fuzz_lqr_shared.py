"""One-off: LQR solves whose F, f, C, c are SHARED by the batch (batch stride 0 in the C ABI) on random shapes, against the
same solve with the operands replicated per instance -- bit for bit (same kernel, same arithmetic).
python tools/probes/fuzz_lqr_shared.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.solvers.lqr import LQR
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(31)
bad = 0
for case in range(cases):
    n, m = int(rng.integers(1, 41)), int(rng.integers(1, 25))
    T, B = int(rng.integers(1, 25)), int(rng.choice([2, 33, 65, 300, 2100]))
    F, f, C, c, _ = problems.make_lqr_batch_fast(1, n, m, seed=case)
    F = F * 0.5 * 2.0 / np.sqrt(n)
    x0 = rng.normal(size=(B, n, 1)).astype(np.float32)
    shared = LQR(F[0], f[0], C[0], c[0]).solve_device(x0, T, want_policy=True); torch.cuda.synchronize()
    rep = LQR(np.repeat(F, B, 0), np.repeat(f, B, 0), np.repeat(C, B, 0), np.repeat(c, B, 0)).solve_device(x0, T, want_policy=True); torch.cuda.synchronize()
    ok = all(torch.equal(shared[k].reshape(rep[k].shape) if shared[k].numel() == rep[k].numel() else shared[k], rep[k]) for k in ("states", "actions", "costs", "status"))
    # the shared solve returns ONE policy (K, k without a batch axis) or a broadcast one: compare values
    Ks, Kr = shared["K"], rep["K"]
    ok = ok and torch.equal(Ks.reshape(-1, *Kr.shape[-3:])[0], Kr[0])
    bad += not ok
    print(f"case {case:3d} n={n:2d} m={m:2d} T={T:2d} B={B:4d} {_hip.load().tfmpc_lqr_kernel_name(n, m, T).decode()[:28]:28s}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("failures:", bad); sys.exit(1 if bad else 0)
