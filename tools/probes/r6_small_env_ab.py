"""res4 / hvac6 at B = 16 384 (tests/workloads.small_env), device time per launch, alternating kernel variants in one process:
    python tools/probes/r6_small_env_ab.py            (TFMPC_COSTATE_COUPLING: None = product, 'runtime' = the run-time shift test of rounds 4-5)
Prints ms per launch (median of 7 blocks of 10 launches) and a hash of every output, so that 'same bits' is checked in the same run."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, workloads
from tfmpc import _hip


def timed(w, reps=10):
    s, out = w["solver"], None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = s.solve_device(w["x0"], w["T"], u_init=w["u0"], workspace=None if out is None else out["workspace"])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def digest(out):
    h = hashlib.sha256()
    for k in ("states", "actions", "costs", "iterations", "status"):
        h.update(out[k].cpu().numpy().tobytes())
    return h.hexdigest()[:12]


for name in sys.argv[1:] or ("res4", "hvac6"):
    w = workloads.small_env(name)
    variants = [None, "runtime"] if name == "res4" and not os.environ.get("AB_ONE") else [None]
    times = {v: [] for v in variants}
    hashes = {}
    for v in variants:
        with _hip.option("TFMPC_COSTATE_COUPLING", v):
            for _ in range(3): timed(w)
    for block in range(7):
        for v in variants:
            with _hip.option("TFMPC_COSTATE_COUPLING", v):
                ms, out = timed(w)
            times[v].append(ms)
            hashes[v] = digest(out)
    for v in variants:
        print(f"{name:6s} coupling={str(v):8s} median {np.median(times[v]):.4f} ms  min {min(times[v]):.4f}  hash {hashes[v]}  kernel {w['solver'].last_kernel}  blocks {' '.join(f'{t:.3f}' for t in times[v])}", flush=True)
