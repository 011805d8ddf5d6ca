"""What the helper teams of the control-limited kernel (ilqr_lq_box_mfma.hip) did in one launch: the board's header and team records read back from the
workspace after a solve of bench.py's stable-open-loop batch (claims, requests answered per team).   python tools/probes/r5_box_board.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "control_limited_stable"
w = getattr(workloads, name)(65536)
s = workloads.solver_of(w)
out = s.solve_device(w["x0"], w["T"], u_init=w["u0"]); torch.cuda.synchronize()
ws = out["workspace"]
B, T = 65536, w["T"]
n, m = int(w["x0"].shape[1]), int(w["u0"].shape[2])
raw = ws.view(torch.uint8)
base = raw.data_ptr()
off = B * T * m * n * 4 + B * T * m * 4
off += (-(base + off)) % 256
hdr = raw[off:off + 256 + 32 * 256].cpu().numpy().view(np.int32)
print(f"{name}: n {n} m {m} T {T}; finished {hdr[0]} claimed {hdr[1]}")
TEAMS = int(os.environ.get('TFMPC_BOX_HELPERS', '8') or 8)      # (the launcher's default; beyond them the slab holds trajectory buffers)
print(f"claims in total {hdr[2]}")
for t in range(TEAMS):
    r = hdr[64 + 64 * t: 64 + 64 * (t + 1)]
    print(f"  team {t}: owner {r[0]} seq {r[1]} instance {r[2]} present {r[3]} done {list(r[4:9])}")
it = out["iterations"].cpu().numpy()
print("iterations: max", it.max(), "instances with >= 8:", int((it >= 8).sum()), "argmax", int(it.argmax()))
