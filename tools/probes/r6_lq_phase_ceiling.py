"""Round 6, VERDICT item 1: the ceiling of a PHASE-SPLIT iLQR.solve on the LQ env at the headline shape (n = 16, m = 8, T = 50, B = 65 536).
Times the existing backward-only (no value outputs: the five-wave instantiation) and forward-only instantiations of lqr_mfma16x8_kernel,
the fused LQR solve and the fused iLQR LQ kernel, each as ONE event pair around `reps` launches after warm-up.  2 x sweep + 2 x rollout is
what a phase-kernel form of the bench's `ilqr_api` workload (two sweeps, start rollout + one line-search rollout) could reach at best."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
from tfmpc.solvers.lqr import LQR

B, n, m, T = int(os.environ.get("B", 65536)), 16, 8, 50
reps, warm = 20, 5
F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=4321)
F = 0.25 * F
lqr = LQR(F, f, C, c)
lib = _hip.require_gpu()
x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
K = torch.empty((B, T, m, n), device="cuda"); k = torch.empty((B, T, m, 1), device="cuda")
status = torch.empty((B,), dtype=torch.int32, device="cuda")
states = torch.empty((B, T + 1, n, 1), device="cuda"); actions = torch.empty((B, T, m, 1), device="cuda"); costs = torch.empty((B, T + 1, 1, 1), device="cuda")


def timed(fn):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def bw():
    _hip.check(lib.tfmpc_lqr_backward_f32(B, n, m, T, *lqr._ptr_args(), _hip.ptr(K), _hip.ptr(k), None, None, None, _hip.ptr(status), _hip.stream()), "bw")


def fw():
    _hip.check(lib.tfmpc_lqr_forward_f32(B, n, m, T, *lqr._ptr_args(), _hip.ptr(K), T * m * n, _hip.ptr(k), T * m, _hip.ptr(x0d),
                                         _hip.ptr(states), _hip.ptr(actions), _hip.ptr(costs), _hip.stream()), "fw")


out = {"B": B}
out["lqr_backward_only_ms"] = timed(bw)
out["lqr_forward_only_ms"] = timed(fw)
ws = [None]
def solve():
    r = lqr.solve_device(x0d, T, workspace=ws[0]); ws[0] = r["workspace"]
out["lqr_solve_ms"] = timed(solve)
opt = lqr.solve_device(x0d, T)["actions"]
gen = torch.Generator(device="cuda").manual_seed(7)
u0 = (opt + 0.05 * opt.abs().amax(dim=(1, 2, 3), keepdim=True) * torch.randn(opt.shape, device="cuda", generator=gen)).contiguous()
s = iLQR(LQEnv(F, f, C, c))
o = [s.solve_device(x0d, T, u_init=u0)]
def il():
    o[0] = s.solve_device(x0d, T, u_init=u0, workspace=o[0]["workspace"])
out["ilqr_lq_fused_ms"] = timed(il)
out["ilqr_iterations"] = float((o[0]["iterations"].double() + 1).sum())
out["kernel"] = lib.tfmpc_ilqr_last_kernel_name().decode()
out["ceiling_2sweep_2rollout_ms"] = 2 * out["lqr_backward_only_ms"] + 2 * out["lqr_forward_only_ms"]
print(json.dumps(out, indent=1))
