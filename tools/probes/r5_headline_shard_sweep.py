"""Headline kernel at the shard sizes of `bench.py --scaling strong`: register budget for four against five resident waves per SIMD
(TFMPC_LQR_WAVES), kernel time from HIP events around the bare C-ABI call (outputs allocated once), the host cost of one
`LQR.solve_device` call, and the wall time per step of a bench-like loop.  Run on the GPU box:
    python tools/probes/r5_headline_shard_sweep.py > gpurun_out/r5_shard_sweep.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import problems
from tfmpc import _hip
from tfmpc.solvers.lqr import LQR

n, m, T = 16, 8, 50
sizes = [int(s) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else
                          "1024,2048,4096,6144,8192,10240,12288,16384,20480,24576,32768,40960,49152,65536".split(","))]
Bmax = max(sizes)
F, f, C, c, x0 = problems.make_lqr_batch_spd(Bmax, n, m, seed=1234)
full = LQR(F, f, C, c)
x0_full = full._prep_x0(x0)
lib = _hip.require_gpu()
rows = []
for B in sizes:
    lqr = LQR(F[:B], f[:B], C[:B], c[:B])
    x0d = x0_full[:B].contiguous()
    ref = None
    row = {"instances": B, "waves_per_simd": B / 1024.0}
    for eu in ("4", "5", None):
        with _hip.option("TFMPC_LQR_WAVES", eu):
            out = lqr.solve_device(x0d, T)
            ws = out["workspace"]
            args = (B, n, m, T, *lqr._ptr_args(), _hip.ptr(x0d), _hip.ptr(out["states"]), _hip.ptr(out["actions"]), _hip.ptr(out["costs"]),
                    None, None, None, None, None, _hip.ptr(out["status"]), _hip.ptr(ws), ws.numel() * 4, _hip.stream())
            for _ in range(30):
                lib.tfmpc_lqr_solve_f32(*args)
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
            for s, e in ev:
                s.record()
                lib.tfmpc_lqr_solve_f32(*args)
                e.record()
            torch.cuda.synchronize()
            ms = np.array([s.elapsed_time(e) for s, e in ev])
            key = "auto" if eu is None else "eu" + eu
            row[key + "_ms_mean"] = float(ms.mean())
            row[key + "_ms_min"] = float(ms.min())
            row[key + "_ms_median"] = float(np.median(ms))
            # back-to-back launches, no events between them: the wall time per launch once the queue is full
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                lib.tfmpc_lqr_solve_f32(*args)
            torch.cuda.synchronize()
            row[key + "_back_to_back_ms"] = (time.perf_counter() - t0) / 200 * 1e3
            snap = (out["states"].clone(), out["actions"].clone(), out["costs"].clone())
            if ref is None:
                ref = snap
            else:
                row[key + "_bit_identical_to_eu4"] = bool(all(torch.equal(a, b) for a, b in zip(ref, snap)))
    # bench-like loop through the Python API (allocations per step, status memset): wall time per step
    ws = None
    for _ in range(20):
        out = lqr.solve_device(x0d, T, workspace=ws); ws = out["workspace"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        out = lqr.solve_device(x0d, T, workspace=ws); ws = out["workspace"]
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    row["api_wall_ms_per_step"] = (time.perf_counter() - t0) / 200 * 1e3
    row["api_host_ms_per_call"] = t_host / 200 * 1e3
    rows.append(row)
    print(json.dumps(row), file=sys.stderr, flush=True)
print(json.dumps({"workload": "random LQR n=16 m=8 T=50", "rows": rows}, indent=1))
