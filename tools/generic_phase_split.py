"""Shape-generic LQR kernel (lqr_generic.hip) at n=32, m=16, T=50, B=8192: fused solve vs the Riccati sweep alone vs
the rollout alone.  Run on the GPU box: python tools/generic_phase_split.py [n m B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.solvers.lqr import LQR, Policy

n, m, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 16, 8192)
T = 50
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1); F *= 0.5
lqr = LQR(F, f, C, c)
x0 = torch.as_tensor(x0[..., None], device="cuda")
out = lqr.solve_device(x0, T, want_policy=True); torch.cuda.synchronize()
K, k, ws = out["K"], out["k"], out["workspace"]
pol = Policy(K, k)
lib = _hip.require_gpu()
status = torch.zeros(B, dtype=torch.int32, device="cuda")
print("kernel:", lib.tfmpc_lqr_kernel_name(n, m, T).decode())


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def sweep():
    rc = lib.tfmpc_lqr_backward_f32(B, n, m, T, *lqr._ptr_args(), _hip.ptr(K), _hip.ptr(k), None, None, None,
                                    _hip.ptr(status), _hip.stream())
    assert rc == 0


print(f"n={n} m={m} B={B} T={T}")
print(f"fused solve   {timeit(lambda: lqr.solve_device(x0, T, workspace=ws)):.2f} ms")
print(f"sweep alone   {timeit(sweep):.2f} ms")
print(f"rollout alone {timeit(lambda: lqr.forward(pol, x0, T)):.2f} ms")
