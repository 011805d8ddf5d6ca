"""bench.py's stable control-limited batch (65 536) under TFMPC_BOX_SPECULATE = argv[1:]: ms per launch and a hash of every output (same bits?)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
from tfmpc import _hip
w = workloads.control_limited_stable(65536)
s = workloads.solver_of(w)
for v in sys.argv[1:] or ("off", "0", "1", "2"):
    with _hip.option("TFMPC_BOX_SPECULATE", v):
        out = None
        for _ in range(3):
            out = s.solve_device(w["x0"], w["T"], u_init=w["u0"], workspace=None if out is None else out["workspace"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = s.solve_device(w["x0"], w["T"], u_init=w["u0"], workspace=out["workspace"])
        e1.record(); torch.cuda.synchronize()
    h = hashlib.sha256()
    for k in ("states", "actions", "costs", "iterations", "status"):
        h.update(out[k].cpu().numpy().tobytes())
    print(f"TFMPC_BOX_SPECULATE={v:4s} {e0.elapsed_time(e1) / 5:7.2f} ms per launch   sha256 {h.hexdigest()[:16]}", flush=True)
