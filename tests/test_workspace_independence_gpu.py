"""No result may depend on what the caller's workspace held before the call.  Every solver entry point takes its scratch
from the caller (include/tfmpc_hip.h: "Scratch is passed in explicitly"), so the same buffer is reused across calls and
shapes; each kernel family is run with the workspace pre-filled with zeros, NaN and +inf and must return identical bits.
(Found with tools/probes/fuzz_costate.py: with several instances per matrix-core column the empty sub-columns of the last
wave read trajectory slots nobody had written, and a stale NaN reached the live rows through 0 x NaN.)"""

import numpy as np
import pytest
import torch

import problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.lq import LQEnv
from tfmpc.envs.lqr.navigation import NavigationLQR
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
from tfmpc.solvers.lqr import LQR

pytestmark = pytest.mark.gpu

KEYS = ("states", "actions", "costs", "iterations", "status")


def _same(a, b):
    return torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))


def _check(solve):
    first = solve(None)
    torch.cuda.synchronize()
    ws = first["workspace"]
    outs = []
    for fill in (0.0, float("nan"), float("inf")):
        ws.fill_(fill)
        out = solve(ws)
        torch.cuda.synchronize()
        outs.append({k: out[k].clone() for k in KEYS if k in out})
    for other in outs[1:]:
        for k in outs[0]:
            assert _same(outs[0][k], other[k]), k


def _costate_env(kind, n):
    cfg = problems.hvac_config(n, seed=1) if kind == "hvac" else problems.reservoir_config(n, seed=1)
    return (HVAC if kind == "hvac" else Reservoir).load(dict(cfg))


@pytest.mark.parametrize("kernel", ["costate_mfma", "lean", "wave"])
@pytest.mark.parametrize("kind,n,B", [("reservoir", 2, 2), ("reservoir", 4, 5), ("reservoir", 7, 3), ("hvac", 3, 2), ("hvac", 6, 33),
                                      ("reservoir", 12, 5), ("reservoir", 20, 5), ("hvac", 32, 17)])
def test_costate_kernels(kernel, kind, n, B):
    env = _costate_env(kind, n)
    x0 = np.random.default_rng(1).uniform(20.0, 60.0, size=(B, n, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=5)
    u0 = solver.random_actions(30, B, seed=3)
    with _hip.option("TFMPC_ILQR_KERNEL", kernel):
        _check(lambda ws: solver.solve_device(x0, 30, u_init=u0, workspace=ws))


@pytest.mark.parametrize("storage_bf16", [False, True])
def test_costate_sixteen_bit_containers(storage_bf16):
    env = _costate_env("reservoir", 6)
    x0 = np.random.default_rng(2).uniform(20.0, 60.0, size=(19, 6, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=4, storage_bf16=storage_bf16)
    u0 = solver.random_actions(20, 19, seed=3)
    _check(lambda ws: solver.solve_device(x0, 20, u_init=u0, workspace=ws))


@pytest.mark.parametrize("B", [1, 5, 67, 2300])
def test_navigation_lane_and_group_kernels(B):
    env = Navigation.load(problems.NAV_CONFIG)
    x0 = np.random.default_rng(3).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=6)
    u0 = solver.random_actions(20, B, seed=4)
    _check(lambda ws: solver.solve_device(x0, 20, u_init=u0, workspace=ws))


@pytest.mark.parametrize("beta,bound", [(0.0, None), (5.0, 1.0)])
def test_navigation_lqr(beta, bound):
    low, high = (None, None) if bound is None else (-bound, bound)
    solver = iLQR(NavigationLQR([[5.5], [-9.0]], beta, low, high), max_iterations=5)
    x0 = np.random.default_rng(5).normal(size=(9, 2, 1)).astype(np.float32)
    u0 = solver.random_actions(12, 9, seed=6)
    _check(lambda ws: solver.solve_device(x0, 12, u_init=u0, workspace=ws))


@pytest.mark.parametrize("n,m,T,bound", [(16, 8, 20, None), (16, 8, 20, 0.5), (9, 3, 12, 1.0), (32, 16, 10, None), (24, 12, 8, None), (40, 8, 6, None)])
def test_lq_env_kernels(n, m, T, bound):
    B = 7
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=n + m)
    low, high = (None, None) if bound is None else (-bound, bound)
    solver = iLQR(LQEnv(F * 0.25 * np.sqrt(16.0 / n), f, C, c, low=low, high=high), max_iterations=4)
    u0 = np.zeros((B, T, m, 1), dtype=np.float32)
    _check(lambda ws: solver.solve_device(x0.astype(np.float32)[..., None], T, u_init=u0, workspace=ws))


@pytest.mark.parametrize("n,m,T,B", [(16, 8, 20, 5), (7, 3, 9, 4), (2, 2, 10, 70), (32, 16, 8, 3), (40, 24, 5, 2), (20, 4, 6, 9)])
def test_lqr_kernels(n, m, T, B):
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=n * 3 + m)
    lqr = LQR(F * 0.5, f, C, c)
    x0d = lqr._prep_x0(x0)
    _check(lambda ws: lqr.solve_device(x0d, T, workspace=ws))


def test_concurrent_streams_do_not_interfere():
    """Solves of different families in flight on different streams (each with its own workspace) return the bits of the same
    solves run one after the other: no kernel keeps state outside the buffers it is handed."""
    rng = np.random.default_rng(9)
    F, f, C, c, x0 = problems.make_lqr_batch_fast(3000, 16, 8, seed=2)
    lqr = LQR(F * 0.4, f, C, c)
    x0d = lqr._prep_x0(x0)
    res = iLQR(_costate_env("reservoir", 20), max_iterations=6)
    xr = rng.uniform(20.0, 60.0, size=(700, 20, 1)).astype(np.float32)
    ur = res.random_actions(40, 700, seed=1)
    nav = iLQR(Navigation.load(problems.NAV_CONFIG), max_iterations=8)
    xn = torch.as_tensor(rng.uniform(0, 10, size=(900, 2, 1)).astype(np.float32), device="cuda")
    un = nav.random_actions(30, 900, seed=2)
    xr = torch.as_tensor(xr, device="cuda")
    jobs = [lambda ws: lqr.solve_device(x0d, 40, workspace=ws),
            lambda ws: res.solve_device(xr, 40, u_init=ur, workspace=ws),
            lambda ws: nav.solve_device(xn, 30, u_init=un, workspace=ws)]
    serial = []
    for job in jobs:
        out = job(None)
        torch.cuda.synchronize()
        serial.append({k: out[k].clone() for k in KEYS if k in out})
    streams = [torch.cuda.Stream() for _ in jobs]
    ws = [None] * len(jobs)
    outs = [None] * len(jobs)
    torch.cuda.synchronize()
    for rep in range(3):
        for i, (job, st) in enumerate(zip(jobs, streams)):
            with torch.cuda.stream(st):
                outs[i] = job(ws[i])
                ws[i] = outs[i]["workspace"]
    torch.cuda.synchronize()
    for ref, out in zip(serial, outs):
        for k in ref:
            assert _same(ref[k], out[k]), k
