"""Control-limited kernel, 65 536 problems: where the sweeps and rollouts of the long-running instances go (probe build:
tools/probes/build_boxprobe.sh: ilqr_lq_box_mfma.hip with -DTFMPC_BOX_PROBE, loaded through TFMPC_LIB).  A rejected line search is followed by a pass
that probes the SAME regularisation levels shifted by one (mu <- max(mu_min, mu delta) is the local bump's own step, ilqr.py:267-270
vs :308-309) on the SAME nominal trajectory: counted here as 'repeats'."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TFMPC_LIB", os.path.join(ROOT, "tools/probes/ab/lib_boxprobe.so"))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
B = 65536
SCALE = float(os.environ.get("BOX_F_SCALE", "0.25"))         # 0.25: bench.py's `control_limited`; 0.18: its stable-open-loop variant (tests/workloads.py)
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, 16, 8, seed=4321)
s = iLQR(LQEnv(SCALE * F, f, C, c, low=-0.5, high=0.5))
u0 = torch.zeros(B, 50, 8, 1, device="cuda")
x0 = x0[..., None].astype(np.float32)
lib = _hip.load()
lib.tfmpc_debug_box_counts.argtypes = [ctypes.c_void_p]
buf = torch.zeros((B, 16), dtype=torch.int32, device="cuda")
assert lib.tfmpc_debug_box_counts(buf.data_ptr()) == 0
out = s.solve_device(x0, 50, u_init=u0); torch.cuda.synchronize()
c = buf.cpu().numpy().astype(np.int64)
it = out["iterations"].cpu().numpy() + 1
st = out["status"].cpu().numpy()
work = c[:, 0] + 0.4 * c[:, 2]                       # a rollout + its cost pass ~ 0.4 sweeps
order = np.argsort(-work)
print("sweeps: total", c[:, 0].sum(), "of which repeats", c[:, 1].sum(), "| rollouts: total", c[:, 2].sum(), "of which repeats", c[:, 3].sum())
for name, sel in (("top 64 by work", order[:64]), ("top 1 %", order[:B // 100]), ("top 10 %", order[:B // 10]), ("all", order)):
    print(f"{name:15s}: sweeps per instance {c[sel, 0].mean():8.1f} (repeats {c[sel, 1].mean():8.1f}), rollouts {c[sel, 2].mean():8.1f} (repeats {c[sel, 3].mean():8.1f}), "
          f"iterations {it[sel].mean():6.1f}, capped {int(((st[sel] & 16) != 0).sum())}")
for name, sel in (("top 1 %", order[:B // 100]), ("top 10 %", order[:B // 10]), ("all", order)):
    print(f"{name:15s}: failed probes per instance {c[sel, 4].mean():8.1f} running {c[sel, 5].sum() / max(c[sel, 4].sum(), 1):5.1f} of 50 steps each; "
          f"successful sweeps {(c[sel, 0] - c[sel, 4]).mean():8.1f}; time steps in failed probes {c[sel, 5].sum() / max(c[sel, 5].sum() + c[sel, 6].sum(), 1):.2f} of all sweep steps")
print("box-QP iterations per sweep step:", c[:, 7].sum() / (c[:, 5].sum() + c[:, 6].sum()), "| top 1 %:", c[order[:B // 100], 7].sum() / (c[order[:B // 100], 5].sum() + c[order[:B // 100], 6].sum()))
steps = c[:, 5].sum() + c[:, 6].sum()
print(f"cycles (s_memtime ticks x 1024): box-QP {c[:, 8].sum()}, whole sweeps {c[:, 9].sum()}, rollouts {c[:, 10].sum()} | per sweep step: {1024 * c[:, 9].sum() / steps:.0f} ticks, of which box-QP {1024 * c[:, 8].sum() / steps:.0f}; per rollout {1024 * c[:, 10].sum() / c[:, 2].sum():.0f}")
print("Armijo trials per box-QP iteration:", c[:, 11].sum() / c[:, 7].sum(), "| per sweep step:", c[:, 11].sum() / steps)
print(f"inside the box-QP, ticks per projected-Newton iteration: gradient / clamp test / system set-up {1024 * c[:, 14].sum() / c[:, 7].sum():.0f}, LDL^T {1024 * c[:, 12].sum() / c[:, 7].sum():.0f}, "
      f"direction + backtracking + broadcast {1024 * c[:, 13].sum() / c[:, 7].sum():.0f} (of {1024 * c[:, 8].sum() / c[:, 7].sum():.0f})")
# list scheduling on 2 048 wave slots: the launch time a block order gives (work = this instance's ticks in sweeps + rollouts)
import heapq
ticks = (c[:, 9] + c[:, 10]).astype(np.float64) * 1024
def makespan(order_):
    slots = [0.0] * 2048
    heapq.heapify(slots)
    end = 0.0
    for b in order_:
        t0 = heapq.heappop(slots); t1 = t0 + ticks[b]; end = max(end, t1); heapq.heappush(slots, t1)
    return end
first = c[:, 15]
print("first-pass regularisation level histogram:", np.bincount(np.clip(first, 0, 41))[:16].tolist())
print("top 1 % by work: first-pass level quantiles", np.quantile(first[order[:B // 100]], [0, 0.1, 0.5, 0.9, 1.0]).tolist())
base = makespan(range(B))
print(f"simulated launch (ticks): index order {base:.3e}; sorted by first-pass level (descending) {makespan(np.argsort(-first, kind='stable')):.3e}; "
      f"sorted by true work {makespan(order):.3e}; work / slots {ticks.sum() / 2048:.3e}; heaviest instance {ticks.max():.3e}")
print("work quantiles (sweep equivalents): p50", np.quantile(work, 0.5), "p90", np.quantile(work, 0.9), "p99", np.quantile(work, 0.99), "p99.9", np.quantile(work, 0.999), "max", work.max())
