#!/usr/bin/env python3
"""Storage-precision sweep of BASELINE configs[4] ("fp32 vs bf16 tolerance sweep": iLQR on HVAC / Reservoir, n = m = 32, T = 100):
trajectories kept in HBM at fp32 vs in REAL 16-bit containers (arithmetic fp32 in both; what is stored and re-read is
/root/reference/tfmpc/solvers/ilqr.py:174-212's states / actions / costs), against the fp64 CPU restatement (oracle/ilqr_ref.py).

Two views: (a) after ONE iteration (continuous dependence on the data, before line-search decisions can diverge): relative state error
against ``ILQRRef(float64)`` on `n_oracle` instances, and the bf16-vs-fp32 state difference over the batch; (b) after 12 iterations:
relative difference of the achieved total cost, bf16 storage vs fp32 storage, over the batch.

``sweep()`` is what tests/test_bf16_storage_sweep_gpu.py asserts bounds on and what bench.py quotes in `extra`;
``python tests/bf16_sweep.py > profiles/rNN_bf16_storage_sweep.json`` writes the table (GPU box)."""

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

import problems


def sweep(n=32, T=100, B=1024, n_oracle=8, kinds=("hvac", "reservoir")):
    from oracle import envs_ref, ilqr_ref
    from tfmpc import _hip
    from tfmpc.envs.hvac import HVAC
    from tfmpc.envs.reservoir import Reservoir
    from tfmpc.solvers.ilqr import iLQR
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import source_stamp
    out = {"config": f"n=m={n}, T={T}, B={B}, default dispatch: 16-instances-per-wave costate kernel, real bf16 trajectory containers; "
                     f"fp64 restatement on {n_oracle} instances",
           "csrc_sha16": source_stamp.stamp(), "envs": {}}
    rng = np.random.default_rng(5)
    for kind in kinds:
        if kind == "hvac":
            cfg = problems.hvac_config(n, seed=5)
            env, oenv = HVAC.load(dict(cfg)), envs_ref.HVAC(**cfg)
            x0 = (10.0 + rng.normal(0, 1.0, size=(B, n, 1))).astype(np.float32)
        else:
            cfg = problems.reservoir_config(n, seed=5)
            env, oenv = Reservoir.load(dict(cfg)), envs_ref.Reservoir(**cfg)
            x0 = rng.uniform(50.0, 75.0, size=(B, n, 1)).astype(np.float32)
        u0 = iLQR(env).random_actions(T, B, seed=5)
        res = {}
        for iters in (1, 12):
            runs = {}
            for mode in ("fp32", "bf16"):
                s = iLQR(env, max_iterations=iters, storage_bf16=(mode == "bf16"))
                o = s.solve_device(x0, T, u_init=u0)
                torch.cuda.synchronize()
                runs[mode] = {k: o[k].double().cpu().numpy() for k in ("states", "actions", "costs")}
                runs[mode]["iterations"] = o["iterations"].cpu().numpy()
            if iters == 1:
                errs = {m: [] for m in runs}
                for b in range(n_oracle):
                    ref = ilqr_ref.ILQRRef(oenv, max_iterations=1)
                    x, u, c, _ = ref.solve(x0[b].astype(np.float64), T, u_init=u0[b].cpu().numpy().astype(np.float64))
                    for mname, r in runs.items():
                        errs[mname].append(float(np.abs(r["states"][b, ..., 0] - x).max() / np.abs(x).max()))
                res["one_iteration_state_rel_err_vs_fp64"] = {m: {"max": max(v), "median": float(np.median(v))} for m, v in errs.items()}
                d = np.abs(runs["bf16"]["states"] - runs["fp32"]["states"]).reshape(B, -1).max(1) / \
                    np.abs(runs["fp32"]["states"]).reshape(B, -1).max(1)
                res["one_iteration_state_rel_diff_bf16_vs_fp32"] = {"median": float(np.median(d)), "p99": float(np.quantile(d, 0.99)),
                                                                    "max": float(d.max())}
            else:
                tf, tb = runs["fp32"]["costs"].sum(1), runs["bf16"]["costs"].sum(1)
                d = (tb - tf) / np.abs(tf)
                res["twelve_iterations_total_cost_rel_diff_bf16_vs_fp32"] = {
                    "median": float(np.median(d)), "p01": float(np.quantile(d, 0.01)), "p99": float(np.quantile(d, 0.99)),
                    "mean_abs": float(np.abs(d).mean())}
                res["twelve_iterations_mean_iterations"] = {m: float(runs[m]["iterations"].mean() + 1) for m in runs}
        out["envs"][kind] = res
    return out


def headline(table):
    """The three numbers bench.py quotes: HVAC one-iteration state error vs fp64 at fp32 and at bf16 storage, and the median
    12-iteration cost difference bf16 vs fp32."""
    h = table["envs"]["hvac"]
    return {"hvac_one_iteration_state_rel_err_vs_fp64_fp32_storage": h["one_iteration_state_rel_err_vs_fp64"]["fp32"]["max"],
            "hvac_one_iteration_state_rel_err_vs_fp64_bf16_storage": h["one_iteration_state_rel_err_vs_fp64"]["bf16"]["max"],
            "hvac_twelve_iterations_total_cost_rel_diff_bf16_vs_fp32_median": h["twelve_iterations_total_cost_rel_diff_bf16_vs_fp32"]["median"]}


if __name__ == "__main__":
    print(json.dumps(sweep(), indent=1))
