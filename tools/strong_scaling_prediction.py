"""Single-GPU kernel times of the headline workload at the per-GPU shard sizes of `bench.py --scaling strong` (global batch 65 536
block-sharded over N = 1, 2, 4, 8 GPUs -> 65 536 / 32 768 / 16 384 / 8 192 instances per GPU): the prediction the strong-scaling curve
the driver measures on an 8-GPU node can be held against (instances are independent and nothing is exchanged during the timed steps, so
the N-GPU step time is the slowest rank's kernel time at its shard size plus the barrier).  Run on the GPU box:
    python tools/strong_scaling_prediction.py > gpurun_out/strong_scaling_prediction.json      (copy to profiles/)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import problems
from tfmpc.solvers.lqr import LQR

n, m, T, GLOBAL = 16, 8, 50, 65536
rows = []
for N in (1, 2, 4, 8):
    B = GLOBAL // N
    F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=1234)
    lqr = LQR(F, f, C, c)
    x0d = lqr._prep_x0(x0)
    out = lqr.solve_device(x0d, T)
    for _ in range(10):
        out = lqr.solve_device(x0d, T, workspace=out["workspace"])
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for s, e in ev:
        s.record()
        out = lqr.solve_device(x0d, T, workspace=out["workspace"])
        e.record()
    torch.cuda.synchronize()
    ms = np.array([s.elapsed_time(e) for s, e in ev])
    rows.append({"n_gpus": N, "instances_per_gpu": B, "kernel_ms_mean": float(ms.mean()), "kernel_ms_min": float(ms.min()),
                 "waves_per_simd": B / 1024.0})
t1 = rows[0]["kernel_ms_mean"]
for r in rows:
    r["predicted_global_iterations_per_s"] = GLOBAL / (r["kernel_ms_mean"] * 1e-3)
    r["predicted_strong_scaling_efficiency"] = t1 / (r["n_gpus"] * r["kernel_ms_mean"])
print(json.dumps({"workload": "random LQR n=16 m=8 T=50 (bench.py headline), one launch per step", "global_batch": GLOBAL,
                  "note": "per-GPU kernel time at the shard size of N ranks, measured on ONE GPU with HIP events over 40 launches; "
                          "prediction = no cross-GPU cost inside the timed steps (there is no collective in them)", "rows": rows}, indent=1))
