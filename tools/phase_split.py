"""Headline shape: time of the rollout alone (tfmpc_lqr_forward_f32 on a stored policy) vs the fused
solve, to see how the 50 sequential steps split between the sweep and the rollout.
Run on the GPU box: python tools/phase_split.py"""
import sys, time
sys.path.insert(0, '/root/repo/tf-mpc_amd'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np, torch, problems
from tfmpc.solvers.lqr import LQR, Policy

B, n, m, T = 65536, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1)
lqr = LQR(F, f, C, c)
x0 = torch.as_tensor(x0[..., None], device="cuda")
out = lqr.solve_device(x0, T, want_policy=True); torch.cuda.synchronize()
pol = Policy(out["K"], out["k"])

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

ws = out["workspace"]
t_solve = timeit(lambda: lqr.solve_device(x0, T, workspace=ws))
t_fwd = timeit(lambda: lqr.forward(pol, x0, T))
from tfmpc import _hip
lib = _hip.require_gpu()
K, k = out["K"], out["k"]
status = torch.zeros(B, dtype=torch.int32, device="cuda")
def bw():
    rc = lib.tfmpc_lqr_backward_f32(B, n, m, T, *lqr._ptr_args(), _hip.ptr(K), _hip.ptr(k), None, None, None,
                                    _hip.ptr(status), _hip.stream())
    assert rc == 0
t_bwd = timeit(bw)
print(f"sweep alone (no value outputs) {t_bwd:.3f} ms")
print(f"fused solve {t_solve:.3f} ms; rollout alone {t_fwd:.3f} ms (includes output allocation)")

# Do a sweep-only launch and a rollout-only launch overlap when they run CONCURRENTLY (two streams)?
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
st1 = ctypes.c_void_p(s1.cuda_stream) if False else None
import ctypes
K2, k2 = K.clone(), k.clone()
states = torch.empty((B, T + 1, n, 1), device="cuda"); actions = torch.empty((B, T, m, 1), device="cuda"); costs = torch.empty((B, T + 1, 1, 1), device="cuda")
def both():
    with torch.cuda.stream(s1):
        rc = lib.tfmpc_lqr_backward_f32(B, n, m, T, *lqr._ptr_args(), _hip.ptr(K), _hip.ptr(k), None, None, None, _hip.ptr(status), ctypes.c_void_p(s1.cuda_stream))
    with torch.cuda.stream(s2):
        rc2 = lib.tfmpc_lqr_forward_f32(B, n, m, T, *lqr._ptr_args(), _hip.ptr(K2), K2[0].numel(), _hip.ptr(k2), k2[0].numel(), _hip.ptr(x0),
                                        _hip.ptr(states), _hip.ptr(actions), _hip.ptr(costs), ctypes.c_void_p(s2.cuda_stream))
def timeit2(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print(f"sweep and rollout launched concurrently on two streams: {timeit2(both):.3f} ms per pair")
