"""Probe: the regularisation level the matrix-core box kernel's FIRST backward pass ends on (the launcher's probe launch leaves it in
the workspace, ilqr_lq_box_mfma.hip: box_order_kernel's input) against the number of Cholesky failures of the fp32 / fp64 restatement's
first pass (profiles/r04_box_family_oracle.json), on the heavy instances of bench.py's control-limited workload."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import workloads
B, T, n, m = 65536, 50, 16, 8
w = workloads.control_limited(B)
out = workloads.solver_of(w).solve_device(w["x0"], T, u_init=w["u0"])
torch.cuda.synchronize()
ws = out["workspace"]
ws32 = ws.view(torch.int32) if ws.dtype != torch.int32 else ws
off_u = B * T * m * n + B * T * m + B * (T + 1) * n          # IlqrWsLayout: K, k, x, then u (= the box kernel's wsq slab)
if ws.element_size() != 4:
    ws32 = ws.view(torch.uint8)[: (ws.numel() * ws.element_size()) // 4 * 4].view(torch.int32)
level = ws32[off_u:off_u + B].cpu().numpy()
print("first-pass level histogram:", np.bincount(np.clip(level, 0, 45))[:14].tolist())
rep = json.load(open(os.path.join(ROOT, "profiles", "r04_box_family_oracle.json")))
for group, entries in rep.items():
    if group == "batch_counts":
        continue
    for e in entries:
        p0 = e["first_passes"][0]
        print(group, e["instance"], "device level", int(level[e["instance"]]), "| fp32 failures", p0["fp32"] and p0["fp32"]["cholesky_failures"],
              "fp64 failures", p0["fp64"] and p0["fp64"]["cholesky_failures"], "| g_norm dev/fp32/fp64", p0["device"]["g_norm"], p0["fp32"] and p0["fp32"]["g_norm"],
              p0["fp64"] and p0["fp64"]["g_norm"])
