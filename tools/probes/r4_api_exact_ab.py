"""A/B: ilqr_lq_mfma_kernel<EXACT> (n = 16, m = 8 as compile-time constants; default at that shape) against the shape-generic form
(TFMPC_ILQR_KERNEL=lq_generic) on bench.py's ilqr_api workloads; alternating launches, outputs must be the same bits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
from tfmpc import _hip
for make in (workloads.ilqr_api_warm, workloads.ilqr_api_cold):
    w = make(65536)
    s = workloads.solver_of(w)
    out = s.solve_device(w["x0"], w["T"], u_init=w["u0"]); torch.cuda.synchronize()
    ts, ref = {None: [], "lq_generic": []}, None
    for rep in range(8):
        for mode in (None, "lq_generic"):
            with _hip.option("TFMPC_ILQR_KERNEL", mode):
                t0 = time.perf_counter(); out = s.solve_device(w["x0"], w["T"], u_init=w["u0"], workspace=out["workspace"]); torch.cuda.synchronize()
                ts[mode].append((time.perf_counter() - t0) * 1e3)
            cur = (out["states"].clone(), out["costs"].clone(), out["iterations"].clone())
            if ref is None: ref = cur
            assert all(torch.equal(a, b) for a, b in zip(ref, cur))
    print(f"{w['version']}: exact {min(ts[None]):.3f} ms (median {np.median(ts[None]):.3f}) | generic {min(ts['lq_generic']):.3f} ms (median {np.median(ts['lq_generic']):.3f}); same bits")
