"""The longest instances of the stable-open-loop control-limited batch, probe build (tools/probes/build_boxprobe.sh): time in sweeps and in line searches
(shader-clock ticks of s_memtime: compare the shares, not the units), with and without helper teams (TFMPC_BOX_HELPERS=off).   python tools/probes/r5_box_chain.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TFMPC_LIB", os.path.join(ROOT, "tools/probes/ab/lib_boxprobe.so"))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
from tfmpc import _hip
w = workloads.control_limited_stable(65536)
s = workloads.solver_of(w)
lib = _hip.load()
lib.tfmpc_debug_box_counts.argtypes = [ctypes.c_void_p]
buf = torch.zeros((65536, 16), dtype=torch.int32, device="cuda")
assert lib.tfmpc_debug_box_counts(buf.data_ptr()) == 0
out = s.solve_device(w["x0"], w["T"], u_init=w["u0"]); torch.cuda.synchronize()
c = buf.cpu().numpy().astype(np.int64)
for b in (22144, 56392, 62458):
    tot = c[b, 9] + c[b, 10]
    print(f"instance {b}: sweeps {c[b, 0]} ({c[b, 9] * 1024 / 1e6:.1f} Mticks = {100.0 * c[b, 9] / tot:.0f} %, {c[b, 9] * 1024 / 1e3 / max(c[b, 0], 1):.0f} kticks each), rollouts {c[b, 2]} "
          f"(line searches {c[b, 10] * 1024 / 1e6:.1f} Mticks = {100.0 * c[b, 10] / tot:.0f} %), box-QP iterations per sweep step {c[b, 7] / max(c[b, 5] + c[b, 6], 1):.2f}")
# where a sweep step of the longest instance goes (shader-clock ticks per time step; the probe's own s_memtime reads included)
for b in (22144, 56392):
    steps = max(c[b, 5] + c[b, 6], 1); qp_it = max(c[b, 7], 1)
    print(f"instance {b}: per sweep step {1024 * c[b, 9] / steps:.0f} ticks = box-QP {1024 * c[b, 8] / steps:.0f} ({c[b, 7] / steps:.2f} iterations of "
          f"{1024 * c[b, 8] / qp_it:.0f}: set-up {1024 * c[b, 14] / qp_it:.0f}, LDL^T {1024 * c[b, 12] / qp_it:.0f}, direction + Armijo + broadcast {1024 * c[b, 13] / qp_it:.0f}; "
          f"{c[b, 11] / qp_it:.2f} Armijo rounds per iteration) + products / value update {1024 * (c[b, 9] - c[b, 8]) / steps:.0f}")
