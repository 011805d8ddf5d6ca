"""Per-phase cycle counts of lqr_block_kernel's Riccati step.  Needs a probe build of the library:
  hipcc ... -DTFMPC_PHASE_PROBE -c tf-mpc_amd/csrc/lqr_block.hip -o tools/probes/ab/lqr_block_probe.o, linked with the other
  objects into tools/probes/ab/lib_probe.so (tools/probes/build_probe.sh).  Run on the GPU box."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TFMPC_LIB"] = f'{root}/tools/probes/ab/lib_probe.so'        # tfmpc/_hip.py loads the probe build instead of the product library
sys.path.insert(0, f'{root}/tf-mpc_amd'); sys.path.insert(0, f'{root}/tests')
import numpy as np, torch, problems
from tfmpc.solvers.lqr import LQR
n, m, B, T = 32, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 50
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1); F *= 0.5
lqr = LQR(F, f, C, c)
out = lqr.solve_device(torch.as_tensor(x0[..., None], device="cuda"), T, want_policy=True); torch.cuda.synchronize()
pc = out["K"].reshape(-1)[:8].cpu().numpy() / T
names = ["[W|F'v]", "[Q|q]", "q, aug", "gauss-jordan", "K,k", "KtQ, const", "[V'|v']", "outputs"]
for nm, v in zip(names, pc): print(f"{nm:16s} {v:10.0f} cycles/step")
print(f"{'total':16s} {pc.sum():10.0f}")
