#!/usr/bin/env python3
"""Diagnostics for the 16-instances-per-wave costate kernel (tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip): deviations from
the wave kernel over a set of shapes, and whole-batch times of BASELINE configs[4] under each kernel.

    python tools/costate_mfma_check.py [--time-only] [--small]
"""
import argparse
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
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc import _hip
from tfmpc.solvers.ilqr import iLQR


def force(name):
    _hip.set_option("TFMPC_ILQR_KERNEL", name)


def make(kind, n, B, seed, dense=False):
    rng = np.random.default_rng(100 + n)
    if kind == "hvac":
        cfg = dict(problems.hvac_config(n, seed=seed))
        if dense:
            cfg["adj"] = np.triu(np.ones((n, n), dtype=bool), 1)
        return HVAC.load(cfg), rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32)
    cfg = dict(problems.reservoir_config(n, seed=seed))
    if dense:
        cfg["downstream"] = rng.uniform(0.0, 0.3, size=(n, n)).astype(np.float32) * (1.0 - np.eye(n, dtype=np.float32))
    return Reservoir.load(cfg), rng.uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)


def compare(kind, n, T, B, iters, dense=False, atol=None):
    env, x0 = make(kind, n, B, n, dense)
    kw = {"max_iterations": iters}
    if atol is not None:
        kw["atol"] = atol
    solver = iLQR(env, **kw)
    u0 = solver.random_actions(T, B, seed=n)
    out = {}
    for kern in ("wave", "costate_mfma"):
        force(kern)
        out[kern] = {k: v.clone() for k, v in solver.solve_device(x0, T, u_init=u0).items() if k not in ("workspace", "batched")}
        torch.cuda.synchronize()
    force(None)
    w, f = out["wave"], out["costate_mfma"]
    same_it = (w["iterations"] == f["iterations"])
    line = {"kind": kind, "n": n, "T": T, "B": B, "iters": iters, "dense": dense,
            "iterations_equal_frac": float(same_it.float().mean()),
            "status_equal": bool(torch.equal(w["status"], f["status"])),
            "max_iter": int(w["iterations"].max())}
    for key in ("states", "actions", "costs"):
        d = (w[key] - f[key]).abs().flatten(1).max(dim=1).values
        scale = w[key].abs().flatten(1).max(dim=1).values.clamp_min(1e-30)
        rel = d / scale
        line[key] = {"bit_equal": bool(torch.equal(w[key], f[key])), "max_rel": float(rel.max()),
                     "median_rel": float(rel.median()), "frac_below_1e-4": float((rel < 1e-4).float().mean())}
    tw, tf_ = w["costs"].sum(dim=1), f["costs"].sum(dim=1)
    line["total_cost_rel"] = float(((tw - tf_).abs() / tw.abs().clamp_min(1e-30)).max())
    line["finite"] = bool(torch.isfinite(f["costs"]).all())
    print(json.dumps(line), flush=True)


def timing(kind, n=32, T=100, B=32768, iters=12, reps=3):
    env, _ = make(kind, n, 1, 5)
    if kind == "hvac":
        x0 = np.full((B, n, 1), 10.0, dtype=np.float32) + np.random.default_rng(5).uniform(-2, 2, size=(B, n, 1)).astype(np.float32)
    else:
        x0 = np.random.default_rng(5).uniform(50.0, 75.0, size=(B, n, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=iters)
    u0 = solver.random_actions(T, B, seed=5)
    res = {"kind": kind, "n": n, "T": T, "B": B, "iters": iters}
    for kern in ("lean", "costate_mfma"):
        force(kern)
        out = solver.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        its = float((out["iterations"].double() + 1).sum())
        res[kern] = {"ms": dt * 1e3, "iterations_per_s": its / dt, "mean_iterations": its / B,
                     "total_cost_mean": float(out["costs"].sum(dim=1).mean())}
    force(None)
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time-only", action="store_true")
    ap.add_argument("--small", action="store_true", help="small-batch timings too")
    args = ap.parse_args()
    if not args.time_only:
        for kind in ("reservoir", "hvac"):
            for (n, T, B) in [(32, 24, 70), (21, 13, 9), (17, 7, 5), (16, 9, 7), (12, 11, 4), (6, 20, 40), (32, 1, 1),
                              (8, 12, 130), (4, 15, 33), (2, 5, 3), (28, 30, 16)]:
                compare(kind, n, T, B, 6)
            compare(kind, 32, 100, 512, 12)
            compare(kind, 8, 25, 203, 30, atol=0.05)
            compare(kind, 27, 12, 50, 5, dense=True)
            compare(kind, 16, 12, 50, 5, dense=True)
    for kind in ("hvac", "reservoir"):
        timing(kind)
    if args.small:
        for kind in ("hvac", "reservoir"):
            for B in (16, 1024, 2048, 4096, 8192, 16384):
                timing(kind, n=32, T=100, B=B, iters=12)
            for B in (4096, 16384, 65536):
                timing(kind, n=6 if kind == "hvac" else 4, T=100, B=B, iters=12)
                timing(kind, n=16, T=100, B=B, iters=12)


if __name__ == "__main__":
    main()
