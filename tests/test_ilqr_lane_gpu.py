"""Lane-per-instance fused iLQR solve (2-D navigation envs): every instance must land where
the wave-per-instance kernel and the oracle land.  Both device paths implement the same
equations; their fp32 summation orders differ, so trajectories agree to fp32 accuracy and
discrete line-search decisions may flip on a few instances of the nonlinear env."""

import os

import numpy as np
import pytest
import torch

import problems
from oracle import envs_ref, ilqr_ref
from tfmpc import _hip
from tfmpc.envs.lqr.navigation import NavigationLQR
from tfmpc.envs.navigation import Navigation
from tfmpc.solvers.ilqr import iLQR

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_kernel():
    def set_(name):
        _hip.set_option("TFMPC_ILQR_KERNEL", name)
    yield set_
    set_(None)


def _solve_both(force_kernel, solver, x0, T, u0):
    out = {}
    for kern in ("lane", "wave"):
        force_kernel(kern)
        out[kern] = solver.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
    return out["lane"], out["wave"]


@pytest.mark.parametrize("beta,bounds", [(0.0, None), (5.0, None), (0.0, (-1.0, 1.0)), (5.0, (-1.0, 1.0))])
def test_lane_equals_wave_on_linear_navigation(force_kernel, beta, bounds):
    rng = np.random.default_rng(11)
    B, T = 150, 10
    goals = rng.uniform(-10, 10, size=(B, 2, 1)).astype(np.float32)
    low, high = bounds if bounds else (None, None)
    solver = iLQR(NavigationLQR(goals, beta, low, high))
    x0 = rng.normal(size=(B, 2, 1)).astype(np.float32)
    u0 = solver.random_actions(T, B, seed=3)
    lane, wave = _solve_both(force_kernel, solver, x0, T, u0)
    assert int((lane["status"] & ~_hip.ST_QP_MAXITER).sum()) == 0
    assert torch.equal(lane["iterations"], wave["iterations"])
    for key in ("states", "actions", "costs"):
        scale = float(wave[key].abs().max())
        assert float((lane[key] - wave[key]).abs().max()) <= 2e-4 * scale, key
    # and against the fp64 oracle on a few instances
    for b in (0, 77, 149):
        o = ilqr_ref.ILQRRef(envs_ref.NavigationLQR(goals[b], beta, low, high))
        x, u, c, it = o.solve(x0[b], T, u_init=u0[b].cpu().numpy())
        assert it == int(lane["iterations"][b])
        assert np.abs(lane["states"][b, ..., 0].cpu().numpy() - x).max() <= 1e-3 * max(np.abs(x).max(), 1.0)
        assert abs(float(lane["costs"][b].sum()) - c.sum()) <= 1e-3 * abs(c.sum())


def test_lane_tracks_wave_and_oracle_on_nonlinear_navigation(force_kernel):
    cfg = problems.NAV_CONFIG
    solver = iLQR(Navigation.load(cfg))
    rng = np.random.default_rng(5)
    B, T = 128, 20
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = solver.random_actions(T, B, seed=9)
    lane, wave = _solve_both(force_kernel, solver, x0, T, u0)
    assert int((lane["status"] & (_hip.ST_NAN | _hip.ST_MAX_ATTEMPTS)).sum()) == 0
    tl, tw = lane["costs"].sum(dim=1), wave["costs"].sum(dim=1)
    rel = ((tl - tw).abs() / tw.abs().clamp_min(1e-3)).cpu().numpy()
    assert np.median(rel) <= 1e-4 and np.quantile(rel, 0.9) <= 2e-2          # a few flipped line searches
    same_iters = float((lane["iterations"] == wave["iterations"]).float().mean())
    assert same_iters >= 0.7
    oenv = envs_ref.Navigation(cfg["goal"], cfg["deceleration"]["center"], cfg["deceleration"]["decay"], cfg["low"],
                               cfg["high"], dtype=np.float32)
    for b in (0, 64, 127):
        o = ilqr_ref.ILQRRef(oenv, dtype=np.float32)
        x, u, c, it = o.solve(x0[b], T, u_init=u0[b].cpu().numpy())
        assert abs(float(tl[b]) - c.sum()) <= 2e-2 * abs(c.sum())
    # returned trajectories obey the env exactly as computed by the (wave) env kernels
    env = solver.env
    st, ac = lane["states"], lane["actions"]
    for t in (0, 9, 19):
        nxt = env.transition(st[:, t], ac[:, t], batch=True)
        assert torch.equal(nxt, st[:, t + 1])


@pytest.mark.parametrize("kind,B,T", [("navlqr", 203, 12), ("navigation", 131, 50)])
def test_group_kernel_equals_per_lane_kernel(force_kernel, kind, B, T):
    """The 16-lanes-per-instance kernel evaluates all line-search step sizes at once and picks
    the first accepted one (ilqr.py:322-353); the one-lane-per-instance kernel tries them in
    order.  Same arithmetic, same decisions: outputs must be bit-identical.  B is not a
    multiple of 4 so the last wave has idle groups."""
    rng = np.random.default_rng(21)
    if kind == "navlqr":
        goals = rng.uniform(-10, 10, size=(B, 2, 1)).astype(np.float32)
        solver = iLQR(NavigationLQR(goals, 5.0, -1.0, 1.0))
        x0 = rng.normal(size=(B, 2, 1)).astype(np.float32)
    else:
        solver = iLQR(Navigation.load(problems.NAV_CONFIG))
        x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = solver.random_actions(T, B, seed=2)
    out = {}
    for kern in ("lane", "lane1"):
        force_kernel(kern)
        out[kern] = solver.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
    g, l = out["lane"], out["lane1"]
    assert torch.equal(g["iterations"], l["iterations"])
    assert torch.equal(g["status"], l["status"])
    for key in ("states", "actions", "costs"):
        assert torch.equal(g[key], l[key]), key


@pytest.mark.parametrize("kind,T", [("navigation", 50), ("navlqr", 12)])
def test_four_groups_per_wave_equal_the_per_lane_kernel(force_kernel, kind, T):
    """Batches beyond 2048 instances run four instances per wavefront (smaller ones a wavefront per instance, covered
    above): bit-identical to the one-lane-per-instance kernel, iterations and status included."""
    rng = np.random.default_rng(33)
    B = 2300                                   # > 2048, and not a multiple of 4
    if kind == "navlqr":
        solver = iLQR(NavigationLQR(rng.uniform(-10, 10, size=(B, 2, 1)).astype(np.float32), 5.0, -1.0, 1.0))
        x0 = rng.normal(size=(B, 2, 1)).astype(np.float32)
    else:
        solver = iLQR(Navigation.load(problems.NAV_CONFIG))
        x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = solver.random_actions(T, B, seed=8)
    out = {}
    for kern in ("lane1", None):
        force_kernel(kern)
        out[kern] = solver.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
    for key in ("iterations", "status", "states", "actions", "costs"):
        assert torch.equal(out[None][key], out["lane1"][key]), key


@pytest.mark.parametrize("B", [300, 2300])
def test_candidates_that_are_not_kept_are_rolled_out_again(force_kernel, B):
    """Round 6: only the first TFMPC_GROUP_STORED (default 4) step sizes of a group's line search keep their candidate trajectory; a pass that adopts
    another one rolls it out once more on the whole group.  With 1 or 2 kept almost every backtracking pass takes that path: every output and the decision trace equal the one-lane-per-instance kernel's / each other's bit for bit, in both the
    one-group-per-wave form (B = 300) and the four-groups form with the instance queue (B = 2 300)."""
    rng = np.random.default_rng(41)
    T = 50
    solver = iLQR(Navigation.load(problems.NAV_CONFIG))
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = solver.random_actions(T, B, seed=9)
    force_kernel("lane1")
    ref = solver.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    force_kernel(None)
    traces, beyond = {}, 0
    for stored in ("1", "2", "3", None):
        with _hip.option("TFMPC_GROUP_STORED", stored):
            out = solver.solve_device(x0, T, u_init=u0, trace_rows=100)
            torch.cuda.synchronize()
        assert solver.last_kernel.startswith("lane_group")
        for key in ("iterations", "status", "states", "actions", "costs"):
            assert torch.equal(out[key], ref[key]), (stored, key)
        traces[stored] = torch.nan_to_num(out["trace"], nan=-7.0).clone()
        assert torch.equal(traces[stored], traces["1"]) and int(out["trace_len"].max()) <= 100
    adopted = traces["1"][..., 5]                                     # alpha_index of every pass (-1 / -7: none)
    assert int((adopted >= 1).sum()) > B and int((adopted >= 4).sum()) > 0       # the replay path was taken, also with the default


def test_instance_queue_of_the_persistent_groups(force_kernel):
    """Round 4: the group kernel's grid is what the chip holds at once (~3 840 wavefronts = 15 360 groups) and a group whose
    instance has finished takes the next one from an atomic queue.  20 011 instances are more than one round of groups: every
    instance must come out exactly as the one-lane-per-instance kernel computes it (bit for bit, iterations and status included),
    whichever group picked it up and whatever its neighbours in the wave were doing; and a second launch on the same workspace
    (the queue counter is reset by the launcher) returns the same."""
    rng = np.random.default_rng(44)
    B, T = 20011, 50
    solver = iLQR(Navigation.load(problems.NAV_CONFIG))
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = solver.random_actions(T, B, seed=9)
    force_kernel("lane1")
    ref = solver.solve_device(x0, T, u_init=u0)
    force_kernel(None)
    out = solver.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    first = {k: out[k].clone() for k in ("iterations", "status", "states", "actions", "costs")}
    again = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
    torch.cuda.synchronize()
    for key in ("iterations", "status", "states", "actions", "costs"):
        assert torch.equal(first[key], ref[key]), key
        assert torch.equal(again[key], ref[key]), key
    assert int(ref["iterations"].max()) >= 30 and int(ref["iterations"].min()) <= 3     # the spread the queue exists for


def test_persistent_grid_follows_the_horizon_of_each_launch(force_kernel):
    """the persistent grid (wavefronts the chip holds at once) depends on the launch's dynamic LDS, i.e. on the horizon
    (~208 T bytes per wavefront).  It used to be frozen at the first launch of the process: a first solve at a long horizon left every
    later T = 50 launch with a third of the wavefronts.  Now cached per (device, LDS bytes): a long-horizon launch FIRST, then T = 50 --
    the second grid must be the larger one, and a T = 50 launch that comes before / after agrees with it."""
    lib = _hip.require_gpu()
    rng = np.random.default_rng(5)
    B = 16384
    solver = iLQR(Navigation.load(problems.NAV_CONFIG), max_iterations=2)
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    grids = {}
    for T in (200, 50, 200, 50):          # 41.8 KB against 10.5 KB of LDS per wavefront (208 T + 32 bytes)
        solver.solve_device(x0, T, u_init=solver.random_actions(T, B, seed=1))
        torch.cuda.synchronize()
        grids.setdefault(T, []).append(int(lib.tfmpc_ilqr_last_group_grid()))
    assert grids[50][0] == grids[50][1] and grids[200][0] == grids[200][1], grids
    assert grids[50][0] >= 2 * grids[200][0] > 0, grids


def test_two_variable_boxqp_closed_form_against_the_restatement():
    """The lane kernels' box-QP for two actions (closed form over the nine candidate active sets, iteration as fall-back;
    reached through tfmpc_boxqp_f32 at m = 2) against oracle/boxqp_ref.py (optimization.py:6-101 restated) on 4 000 random
    strictly convex QPs -- minimiser in the interior, on edges, in corners -- plus degenerate ones: a multiplier that is
    exactly zero on a bound, a zero-width box, a nearly singular H."""
    from oracle import boxqp_ref
    from tfmpc.utils import optimization
    rng = np.random.default_rng(21)
    B = 4000
    A = rng.normal(size=(B, 2, 2))
    H = A @ A.transpose(0, 2, 1) + 0.05 * np.eye(2)
    H[:50] = np.array([[1.0, 0.999], [0.999, 1.0]])                       # nearly singular
    q = rng.normal(scale=3.0, size=(B, 2, 1))
    low = -np.abs(rng.normal(size=(B, 2, 1))) - 0.1
    high = np.abs(rng.normal(size=(B, 2, 1))) + 0.1
    low[100:150, 1] = high[100:150, 1] = 0.3                               # zero-width box in the second variable
    # exactly zero multiplier: the unconstrained minimiser sits ON the upper bound of the first variable
    xs = np.linalg.solve(H[200:260], -q[200:260])
    high[200:260, 0] = xs[:, 0]
    low[200:260, 0] = xs[:, 0] - 1.0
    x0 = (low + high) / 2
    x, _, free, clamped = optimization.projected_newton_qp(H, q, low, high, x0)
    x, free = x.cpu().numpy()[..., 0], free.cpu().numpy()[..., 0]
    bad = 0
    for b in range(B):
        xr, _, fr, _ = boxqp_ref.projected_newton_qp(H[b], q[b], low[b], high[b], x0[b], dtype=np.float64)
        scale = max(1.0, np.abs(xr).max())
        assert np.abs(x[b] - xr[:, 0]).max() <= 2e-4 * scale, (b, x[b], xr[:, 0])
        bad += int(not np.array_equal(free[b], fr[:, 0]))
    assert bad <= B // 200, bad          # the free set may differ only where a multiplier or a distance is at the 1e-6 tolerance
