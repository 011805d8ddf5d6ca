#!/usr/bin/env python3
"""Storage-precision sweep of BASELINE configs[4] (iLQR on HVAC / Reservoir, n = m = 32, T = 100):
trajectories and gains kept in HBM at fp32 vs rounded to bf16 on every store (arithmetic fp32 in
both), against the fp64 CPU restatement.  Run on the GPU box:  python tests/bf16_sweep.py > profiles/rNN_bf16_sweep.json

Two views: (a) after ONE iteration (continuous dependence on the data, before line-search decisions
can diverge): relative state error vs fp64; (b) after 12 iterations: relative difference of the
achieved total cost, bf16-storage vs fp32-storage, over the batch."""

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np
import torch

import problems
from oracle import envs_ref, ilqr_ref
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

n, T, B = 32, 100, 1024
out = {"config": f"n=m={n}, T={T}, B={B}, default dispatch: 16-instances-per-wave costate kernel, real bf16 trajectory containers", "envs": {}}
rng = np.random.default_rng(5)
for kind in ("hvac", "reservoir"):
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
        if iters == 1:
            errs = {m: [] for m in runs}
            for b in (0, 1):
                ref = ilqr_ref.ILQRRef(oenv, max_iterations=1)
                x, u, c, _ = ref.solve(x0[b].astype(np.float64), T, u_init=u0[b].cpu().numpy().astype(np.float64))
                for mname, r in runs.items():
                    errs[mname].append(float(np.abs(r["states"][b, ..., 0] - x).max() / np.abs(x).max()))
            res["one_iteration_state_rel_err_vs_fp64"] = {m: max(v) for m, v in errs.items()}
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
    out["envs"][kind] = res
print(json.dumps(out, indent=1))
