#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload: inputs start in pinned HOST
memory, results end in pinned host memory.  Never bench.py's `value` (that one has the inputs
resident in HBM).  Run on the GPU box: python tools/pcie_inclusive.py"""

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np
import torch

import problems
from tfmpc.solvers.lqr import LQR

B, n, m, T = 65536, 16, 8, 50
F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1)
host_in = [torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).pin_memory() for a in (F, f, C, c, x0)]
host_out = {k: torch.empty(s, dtype=torch.float32).pin_memory()
            for k, s in (("states", (B, T + 1, n, 1)), ("actions", (B, T, m, 1)), ("costs", (B, T + 1, 1, 1)))}
ws = None


def step():
    global ws
    dev = [t.to("cuda", non_blocking=True) for t in host_in]
    out = LQR(*dev[:4]).solve_device(dev[4], T, workspace=ws)
    ws = out["workspace"]
    for k in host_out:
        host_out[k].copy_(out[k], non_blocking=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
bytes_moved = sum(t.numel() for t in host_in) * 4 + sum(t.numel() for t in host_out.values()) * 4
print(json.dumps({"workload": f"random LQR n={n} m={m} T={T} B={B}, host->device->solve->host, pinned memory",
                  "ms_per_batch": dt * 1e3, "solves_per_s": B / dt, "host_bytes_per_batch": bytes_moved,
                  "effective_pcie_GBps": bytes_moved / dt / 1e9}))
