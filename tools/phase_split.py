"""Headline shape (random LQR n=16, m=8, T=50, B=65536): how the 50 sequential steps split between the
Riccati sweep and the rollout.  Times the fused solve, the sweep alone (tfmpc_lqr_backward_f32 without
value outputs), the rollout alone (tfmpc_lqr_forward_f32 on the stored policy), and the two launched
concurrently on two streams.  Run on the GPU box: python tools/phase_split.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.solvers.lqr import LQR, Policy

B, n, m, T = 65536, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1)
lqr = LQR(F, f, C, c)
x0 = torch.as_tensor(x0[..., None], device="cuda")
out = lqr.solve_device(x0, T, want_policy=True); torch.cuda.synchronize()
K, k, ws = out["K"], out["k"], out["workspace"]
pol = Policy(K, k)
lib = _hip.require_gpu()
status = torch.zeros(B, dtype=torch.int32, device="cuda")


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def sweep(stream=None):
    rc = lib.tfmpc_lqr_backward_f32(B, n, m, T, *lqr._ptr_args(), _hip.ptr(K), _hip.ptr(k), None, None, None,
                                    _hip.ptr(status), stream if stream is not None else _hip.stream())
    assert rc == 0


print(f"fused solve            {timeit(lambda: lqr.solve_device(x0, T, workspace=ws)):.3f} ms")
print(f"sweep alone            {timeit(sweep):.3f} ms (no value outputs)")
print(f"rollout alone          {timeit(lambda: lqr.forward(pol, x0, T)):.3f} ms (includes output allocation)")

# the two phases launched concurrently: do a sweep launch and a rollout launch overlap?
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
K2, k2 = K.clone(), k.clone()
states = torch.empty((B, T + 1, n, 1), device="cuda"); actions = torch.empty((B, T, m, 1), device="cuda")
costs = torch.empty((B, T + 1, 1, 1), device="cuda")


def both():
    sweep(ctypes.c_void_p(s1.cuda_stream))
    rc = lib.tfmpc_lqr_forward_f32(B, n, m, T, *lqr._ptr_args(), _hip.ptr(K2), K2[0].numel(), _hip.ptr(k2), k2[0].numel(),
                                   _hip.ptr(x0), _hip.ptr(states), _hip.ptr(actions), _hip.ptr(costs), ctypes.c_void_p(s2.cuda_stream))
    assert rc == 0


print(f"sweep + rollout on two streams {timeit(both):.3f} ms per pair")
