"""Register-resident HVAC / Reservoir solve (tf-mpc_amd/csrc/ilqr_adjoint.hip, n <= 32; for small n several
instances per wavefront, `TFMPC_ILQR_KERNEL=lean`; `lean1` forces one) against the generic wave-per-instance kernel.  All implement ilqr.py:214-355 on the bang-bang branch with the same
operation order, so every output must be BIT-identical -- also on Reservoir, where any rounding
difference would flip line-search decisions and change trajectories completely."""

import os

import numpy as np
import pytest
import torch

import problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_kernel():
    def set_(name):
        _hip.set_option("TFMPC_ILQR_KERNEL", name)
    yield set_
    set_(None)


@pytest.mark.parametrize("kind", ["hvac", "reservoir"])
@pytest.mark.parametrize("n,T,B", [(32, 24, 70), (21, 13, 9), (17, 7, 5), (30, 40, 33), (32, 1, 1), (18, 2, 3),
                                   (16, 9, 7), (12, 11, 4), (6, 20, 40), (4, 15, 33), (2, 5, 3), (3, 1, 2), (8, 12, 130),
                                   (5, 30, 257), (7, 3, 1)])
def test_register_resident_kernel_equals_wave_kernel(force_kernel, kind, n, T, B):
    rng = np.random.default_rng(100 + n)
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=n)))
        x0 = rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=n)))
        x0 = rng.uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=6)
    u0 = solver.random_actions(T, B, seed=n)
    out = {}
    for kern in ("lean", "lean1", "wave"):    # packed where the shape allows it / one instance per wave / generic
        force_kernel(kern)
        out[kern] = solver.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
    wave = out["wave"]
    for kern in ("lean", "lean1"):
        fast = out[kern]
        assert torch.equal(fast["iterations"], wave["iterations"]), kern
        assert torch.equal(fast["status"], wave["status"]), kern
        for key in ("states", "actions", "costs"):
            assert torch.equal(fast[key], wave[key]), (kern, key)
        assert bool(torch.isfinite(fast["costs"]).all())



@pytest.mark.parametrize("kind,n", [("hvac", 6), ("hvac", 4), ("reservoir", 4), ("reservoir", 8)])
def test_packed_instances_finish_at_different_times(force_kernel, kind, n):
    """Several instances share a wavefront in lockstep; groups that converge early (loose atol) keep executing with
    masked stores while their neighbours iterate on.  Must still equal the wave kernel bit for bit."""
    B, T = 203, 25
    rng = np.random.default_rng(n)
    if kind == "hvac":
        env = HVAC.load(dict(problems.hvac_config(n, seed=n)))
        x0 = rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32)
    else:
        env = Reservoir.load(dict(problems.reservoir_config(n, seed=n)))
        x0 = rng.uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=30, atol=0.05)
    u0 = solver.random_actions(T, B, seed=n)
    out = {}
    for kern in ("lean", "wave"):
        force_kernel(kern)
        out[kern] = solver.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
    its = out["wave"]["iterations"]
    assert len(torch.unique(its)) >= 5, torch.unique(its)          # the scenario does spread the finishing times
    assert torch.equal(out["lean"]["iterations"], its) and torch.equal(out["lean"]["status"], out["wave"]["status"])
    for key in ("states", "actions", "costs"):
        assert torch.equal(out["lean"][key], out["wave"][key]), key


@pytest.mark.parametrize("n,B", [(4, 50), (8, 21), (13, 9), (16, 6), (27, 5)])
def test_dense_coupling_matrices_sum_in_the_same_order(force_kernel, n, B):
    """The recipes above give chain topologies (one non-zero per column of `downstream`), for which any summation
    order is exact.  With DENSE random couplings the register-resident kernels only stay bit-identical to the wave kernel
    if both split and order the row sums the same way."""
    T = 12
    rng = np.random.default_rng(50 + n)
    cfg = dict(problems.reservoir_config(n, seed=n))
    cfg["downstream"] = rng.uniform(0.0, 0.3, size=(n, n)).astype(np.float32) * (1.0 - np.eye(n, dtype=np.float32))
    hcfg = dict(problems.hvac_config(n, seed=n))
    hcfg["adj"] = np.triu(np.ones((n, n), dtype=bool), 1)                        # every pair of rooms shares a wall
    for env, x0 in ((Reservoir.load(cfg), rng.uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)),
                    (HVAC.load(hcfg), rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32))):
        solver = iLQR(env, max_iterations=5)
        u0 = solver.random_actions(T, B, seed=n)
        out = {}
        for kern in ("lean", "lean1", "wave"):
            force_kernel(kern)
            out[kern] = solver.solve_device(x0, T, u_init=u0)
            torch.cuda.synchronize()
        for kern in ("lean", "lean1"):
            for key in ("states", "actions", "costs", "iterations", "status"):
                assert torch.equal(out[kern][key], out["wave"][key]), (type(env).__name__, kern, key)


def test_per_instance_parameters(force_kernel):
    """Parameters with a batch stride (every instance its own env) go through the same loads."""
    n, T, B = 24, 10, 6
    cfgs = [problems.hvac_config(n, seed=40 + b) for b in range(B)]
    env = HVAC.load(dict(cfgs[0]))
    envs = [HVAC.load(dict(c)) for c in cfgs]
    rng = np.random.default_rng(3)
    x0 = rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32)
    u0 = iLQR(env).random_actions(T, B, seed=1)
    for b in range(B):
        single = iLQR(envs[b], max_iterations=4)
        force_kernel("lean")
        fast = single.solve_device(x0[b:b + 1], T, u_init=u0[b:b + 1])
        force_kernel("wave")
        wave = single.solve_device(x0[b:b + 1], T, u_init=u0[b:b + 1])
        torch.cuda.synchronize()
        for key in ("states", "actions", "costs"):
            assert torch.equal(fast[key], wave[key]), (b, key)


def test_first_iterations_match_the_fp64_oracle_at_n24():
    """Against the oracle (numpy / torch-autodiff restatement of ilqr.py + hvac/__init__.py), fp64, at a
    size the register-resident kernel serves.  HVAC is smooth enough that two iterations agree to fp32
    accuracy: same line-search decisions, trajectories within 1e-4 relative."""
    from oracle import envs_ref, ilqr_ref
    n, T = 24, 12
    cfg = problems.hvac_config(n, seed=7)
    rng = np.random.default_rng(1)
    x0 = rng.uniform(5.0, 30.0, size=(n, 1)).astype(np.float32)
    solver = iLQR(HVAC.load(dict(cfg)), max_iterations=2)
    u0 = solver.random_actions(T, None, seed=3)
    traj, iteration = solver.solve(x0, T, show_progress=False, u_init=u0)
    o = ilqr_ref.ILQRRef(envs_ref.HVAC(**cfg, dtype=np.float64), dtype=np.float64, max_iterations=2)
    xs, us, cs, it64 = o.solve(x0.astype(np.float64), T, u_init=u0.cpu().numpy().astype(np.float64))
    assert iteration == it64
    scale = np.abs(xs).max()
    assert np.abs(traj.states - xs).max() <= 1e-4 * scale
    assert np.abs(traj.actions - us).max() <= 1e-4
    assert abs(traj.total_cost - cs.sum()) <= 1e-4 * abs(cs.sum())
