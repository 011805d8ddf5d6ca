"""Single-GPU times of the headline workload at the per-GPU shard sizes of `bench.py --scaling strong` (global batch 65 536
block-sharded over N = 1, 2, 4, 8 GPUs -> 65 536 / 32 768 / 16 384 / 8 192 instances per GPU): the prediction the strong-scaling curve
the driver measures on an 8-GPU node can be held against (instances are independent and nothing is exchanged during the timed steps, so
the N-GPU step time is the slowest rank's step time at its shard size plus the barrier).  Two clocks per shard size:
  kernel_ms     HIP events around the bare C-ABI call (outputs allocated once), mean / min over 60 launches;
  step_wall_ms  wall time per step of `bench.py`'s own loop (LQR.solve_device back to back, one synchronisation at the end) --
                this is what `ms_per_step` of an N-GPU run is made of, and what the efficiency column uses.
Run on the GPU box:
    python tools/strong_scaling_prediction.py > gpurun_out/strong_scaling_prediction.json      (copy to profiles/)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import problems
from tfmpc import _hip
from tfmpc.solvers.lqr import LQR

n, m, T, GLOBAL = 16, 8, 50, 65536
lib = _hip.require_gpu()
rows = []
for N in (1, 2, 4, 8):
    B = GLOBAL // N
    F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=1234)
    lqr = LQR(F, f, C, c)
    x0d = lqr._prep_x0(x0)
    out = lqr.solve_device(x0d, T)
    ws = out["workspace"]
    args = (B, n, m, T, *lqr._ptr_args(), _hip.ptr(x0d), _hip.ptr(out["states"]), _hip.ptr(out["actions"]), _hip.ptr(out["costs"]),
            None, None, None, None, None, _hip.ptr(out["status"]), _hip.ptr(ws), ws.numel() * 4, _hip.stream())
    for _ in range(40):
        lib.tfmpc_lqr_solve_f32(*args)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for s, e in ev:
        s.record()
        lib.tfmpc_lqr_solve_f32(*args)
        e.record()
    torch.cuda.synchronize()
    ms = np.array([s.elapsed_time(e) for s, e in ev])
    steps = max(40, int(0.2 / (ms.mean() * 1e-3)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = lqr.solve_device(x0d, T, workspace=ws)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    rows.append({"n_gpus": N, "instances_per_gpu": B, "waves_per_simd": B / 1024.0, "kernel_ms_mean": float(ms.mean()),
                 "kernel_ms_min": float(ms.min()), "step_wall_ms": wall, "steps_timed": steps})
t1 = rows[0]["step_wall_ms"]
k1 = rows[0]["kernel_ms_mean"]
for r in rows:
    r["predicted_global_iterations_per_s"] = GLOBAL / (r["step_wall_ms"] * 1e-3)
    r["predicted_strong_scaling_efficiency"] = t1 / (r["n_gpus"] * r["step_wall_ms"])
    r["kernel_only_efficiency"] = k1 / (r["n_gpus"] * r["kernel_ms_mean"])
print(json.dumps({"workload": "random LQR n=16 m=8 T=50 (bench.py headline), one launch per step", "global_batch": GLOBAL,
                  "note": "per-GPU times at the shard size of N ranks, measured on ONE GPU; prediction = no cross-GPU cost inside the "
                          "timed steps (there is no collective in them)", "rows": rows}, indent=1))
