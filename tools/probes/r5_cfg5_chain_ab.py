"""A/B: Reservoir n = 32, T = 100, 12 iterations through the compile-time chain instantiation (default when the env states its topology,
TfmpcEnv.coupling_shift) against the general two-tile kernel with the run-time shift test (TFMPC_COSTATE_COUPLING=runtime); alternating
launches on one box, bits compared.  B from argv (default 32768 8192 65536)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
from tfmpc import _hip
from tfmpc.solvers.ilqr import iLQR
for B in [int(v) for v in sys.argv[1:]] or (32768, 8192, 65536):
    w = workloads.cfg5("reservoir", B)
    s = iLQR(w["env"], max_iterations=12)
    out = s.solve_device(w["x0"], w["T"], u_init=w["u0"]); torch.cuda.synchronize()
    ts = {None: [], "runtime": []}
    ref = None
    same = True
    for rep in range(6):
        for mode in (None, "runtime"):
            with _hip.option("TFMPC_COSTATE_COUPLING", mode):
                t0 = time.perf_counter(); out = s.solve_device(w["x0"], w["T"], u_init=w["u0"], workspace=out["workspace"]); torch.cuda.synchronize()
                ts[mode].append((time.perf_counter() - t0) * 1e3)
            cur = (out["states"].clone(), out["iterations"].clone(), out["costs"].clone())
            if ref is None: ref = cur
            same = same and all(torch.equal(a, b) for a, b in zip(ref, cur))
    print(f"B={B}: chain instantiation {min(ts[None]):.2f} ms (median {np.median(ts[None]):.2f}) | general kernel, run-time shift {min(ts['runtime']):.2f} ms "
          f"(median {np.median(ts['runtime']):.2f}); same bits: {same}; status {int((out['status'] != 0).sum())}", flush=True)
