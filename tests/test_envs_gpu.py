"""The reference's own env tests, ported in intent onto the device env kernels (``-m gpu``):
``tests/test_diffenv.py:18-70`` (batched call == per-instance calls, shapes, costs >= 0) and the closed
forms of ``tests/test_env_hvac.py:18-211``, ``tests/test_env_reservoir.py:18-233``,
``tests/test_env_navigation.py:6-176``, ``tests/test_env_lqr_navigation.py:28-135``.  The expected values
are numpy expressions of the env parameters written here, not oracle calls.  Tolerance: 1e-5 of each
tensor's scale (the reference's tests use 1e-3 absolute)."""

import numpy as np
import pytest
import torch

import problems
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.lqr.navigation import NavigationLQR
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _close(got, ref, what, rtol=1e-5):
    got, ref = _np(got) if torch.is_tensor(got) else np.asarray(got, float), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.abs(got - ref).max() <= rtol * max(np.abs(ref).max(), 1.0), (what, np.abs(got - ref).max())


def _navigation(zones, seed=0):
    rng = np.random.default_rng(seed)
    return Navigation(np.array([[8.0], [9.0]]), {"center": rng.normal(size=(zones, 2, 1)), "decay": rng.uniform(0.0, 3.0, size=zones)},
                      [[-1.0], [-1.0]], [[1.0], [1.0]])


ENVS = {
    "hvac": lambda: HVAC(**problems.hvac_config(5, seed=3)),
    "reservoir": lambda: Reservoir(**problems.reservoir_config(4, seed=2)),
    "navigation_1_zone": lambda: _navigation(1),
    "navigation_2_zones": lambda: _navigation(2),
    "navigation_lqr": lambda: NavigationLQR([[5.5], [-9.0]], 5.0, -1.0, 1.0),
}


def _sample(env, batch, seed=0):            # reference tests/conftest.py:10-19
    rng = np.random.default_rng(seed)
    x = rng.normal(10.0, 1.0, size=(batch, env.state_size, 1)).astype(np.float32)
    u = rng.uniform(size=(batch, env.action_size, 1)).astype(np.float32)
    return x, u


# ---------------------------------------------------------------- test_diffenv.py ---
@pytest.mark.parametrize("name", list(ENVS))
@pytest.mark.parametrize("batch", [1, 10])
def test_batched_calls_equal_per_instance_calls(name, batch):
    env = ENVS[name]()
    n, m = env.state_size, env.action_size
    x, u = _sample(env, batch)
    nxt = env.transition(x, u, batch=True)
    cost = env.cost(x, u, batch=True)
    assert tuple(nxt.shape) == (batch, n, 1) and tuple(cost.shape) == (batch,)
    if name != "navigation_lqr":
        assert bool((cost >= 0).all())                                              # test_diffenv.py:36
    lin = env.get_linear_transition(x, u, batch=True)
    quad = env.get_quadratic_cost(x, u, batch=True)
    assert len(lin) == 3 and len(quad) == 7
    for i in range(batch):
        assert torch.equal(env.transition(x[i], u[i], batch=False), nxt[i])
        assert torch.equal(env.cost(x[i], u[i], batch=False), cost[i])
        for whole, one in zip(lin, env.get_linear_transition(x[i], u[i], batch=False)):
            assert whole.shape[0] == batch and whole[i].shape == one.shape and torch.equal(whole[i], one)
        for whole, one in zip(quad, env.get_quadratic_cost(x[i], u[i], batch=False)):
            assert whole.shape[0] == batch and whole[i].shape == one.shape and torch.equal(whole[i], one)
    assert tuple(lin.f.shape) == (batch, n, 1) and tuple(lin.f_x.shape) == (batch, n, n) and tuple(lin.f_u.shape) == (batch, n, m)
    assert torch.equal(lin.f, nxt) and torch.equal(quad.l, cost)                      # test_env_hvac.py:55-58,176
    fin = env.get_quadratic_final_cost(x[0])
    assert tuple(fin.l.shape) == () and tuple(fin.l_x.shape) == (n, 1) and tuple(fin.l_xx.shape) == (n, n)
    assert float(fin.l) == pytest.approx(float(env.final_cost(x[0])), rel=1e-6)


# ------------------------------------------------------------------ test_env_hvac.py ---
def test_hvac_parameters_and_closed_forms():
    env = ENVS["hvac"]()
    n = env.state_size
    assert env.temp_lower_bound.shape == (n, 1) and env.temp_upper_bound.shape == (n, 1)       # :18-21
    assert np.all(env.temp_lower_bound <= env.temp_upper_bound)
    assert env.R_outside.shape == (n, 1) and env.R_hall.shape == (n, 1) and env.R_wall.shape == (n, n)   # :24-31
    assert np.all(env.R_outside > 0) and np.all(env.R_hall > 0) and np.all(env.R_wall > 0)
    assert env.adj.shape == (n, n) and not np.tril(env.adj).any()                               # :34-38
    assert "HVAC" in repr(env) and str(env)
    rng = np.random.default_rng(0)
    x = rng.normal(20.0, 6.0, size=(6, n, 1)).astype(np.float32)
    u = rng.uniform(size=(6, n, 1)).astype(np.float32)
    lin = env.get_linear_transition(x, u, batch=True)
    quad = env.get_quadratic_cost(x, u, batch=True)
    cap, air_max, lo, hi = (a.astype(np.float64) for a in (env.capacity, env.air_max, env.temp_lower_bound, env.temp_upper_bound))
    A = np.logical_or(env.adj, env.adj.T).astype(float)
    G = A / env.R_wall
    k_out, k_hall = env.adj_outside / env.R_outside, env.adj_hall / env.R_hall
    for t in range(len(x)):
        xt, ut = x[t].astype(np.float64), u[t].astype(np.float64)
        heating = ut * air_max * env.CAP_AIR * (env.TEMP_AIR - xt)
        rooms = (G * (xt.T - xt)).sum(axis=1, keepdims=True)                                   # :101-118
        nxt = xt + env.TIME_DELTA / cap * (heating + rooms + k_out * (env.temp_outside - xt) + k_hall * (env.temp_hall - xt))
        _close(lin.f[t], nxt, "hvac.f")
        _close(lin.f_u[t], np.diag((env.TIME_DELTA / cap * air_max * env.CAP_AIR * (env.TEMP_AIR - xt))[:, 0]), "hvac.f_u")   # :92-98
        fx = np.eye(n) + np.diag((env.TIME_DELTA / cap)[:, 0]) @ (
            -np.diag((ut * air_max * env.CAP_AIR)[:, 0]) + G - np.diag(G.sum(1)) - np.diag(k_out[:, 0]) - np.diag(k_hall[:, 0]))
        _close(lin.f_x[t], fx, "hvac.f_x")
        lx = env.PENALTY * (-(lo > xt).astype(float) + (xt > hi).astype(float)) - env.SET_POINT_PENALTY * np.sign((lo + hi) / 2 - xt)
        _close(quad.l_x[t], lx, "hvac.l_x")                                                    # :181-203
        _close(quad.l_u[t], air_max * env.COST_AIR, "hvac.l_u")
        for h in ("l_xx", "l_uu", "l_ux", "l_xu"):                                             # :208-211
            assert not _np(getattr(quad, h)[t]).any()
    assert float(env.cost(x[0], u[0])) >= 0 and float(env.final_cost(x[0])) >= 0                # :153-167


# ------------------------------------------------------------- test_env_reservoir.py ---
def test_reservoir_parameters_and_closed_forms():
    env = ENVS["reservoir"]()
    n = env.state_size
    assert env.lower_bound.shape == (n, 1) and env.upper_bound.shape == (n, 1)                  # :18-22
    assert np.all(env.lower_bound < env.upper_bound)
    assert env.downstream.shape == (n, n) and np.all(env.downstream.sum(axis=1) <= 1)           # :25-28
    assert "Reservoir" in repr(env) and str(env)
    rng = np.random.default_rng(0)
    x = rng.uniform(5.0, 95.0, size=(6, n, 1)).astype(np.float32)
    u = rng.uniform(size=(6, n, 1)).astype(np.float32)
    lin = env.get_linear_transition(x, u, batch=True)
    quad = env.get_quadratic_cost(x, u, batch=True)
    D, cap = env.downstream.astype(np.float64), env.max_res_cap.astype(np.float64)
    lo, hi = env.lower_bound.astype(np.float64), env.upper_bound.astype(np.float64)
    rain = env.rain_shape.astype(np.float64) * env.rain_scale
    for t in range(len(x)):
        xt, ut = x[t].astype(np.float64), u[t].astype(np.float64)
        out = ut * xt                                                                          # :105-123
        vap = 0.5 * np.sin(xt / cap) * xt                                                      # :36-52
        nxt = xt + rain - vap - out + D.T @ out                                                # :64-100,126-140
        _close(lin.f[t], nxt, "reservoir.f")
        assert float(_np(lin.f[t]).sum()) == pytest.approx((xt + rain - vap).sum() - out[-1, 0], rel=1e-5)   # mass balance
        fx = (np.eye(n) - np.diag((0.5 * (np.cos(xt / cap) * xt / cap + np.sin(xt / cap)))[:, 0]) - np.diag(ut[:, 0])
              + D.T @ np.diag(ut[:, 0]))
        _close(lin.f_x[t], fx, "reservoir.f_x")                                                # :151-177
        _close(lin.f_u[t], -np.diag(xt[:, 0]) + D.T @ np.diag(xt[:, 0]), "reservoir.f_u")
        LP, HP, SP = -env.low_penalty.astype(np.float64), -env.high_penalty.astype(np.float64), -env.set_point_penalty.astype(np.float64)
        lx = -LP * (lo > xt) + HP * (xt > hi) - SP * np.sign((lo + hi) / 2 - xt)
        _close(quad.l_x[t], lx, "reservoir.l_x")                                               # :189-233
        assert not _np(quad.l_u[t]).any()
        for h in ("l_xx", "l_uu", "l_ux", "l_xu"):
            assert not _np(getattr(quad, h)[t]).any()
    # probabilistic transition (:143-148): cec=False adds rainfall noise, same shape, differs from the mean model
    env.seed(7)
    noisy = env.transition(x[0], u[0], batch=False, cec=False)
    assert noisy.shape == lin.f[0].shape and not torch.equal(noisy, lin.f[0])


def test_reservoir_sine_and_cosine_over_both_paths_of_trig_h():
    """csrc/trig.h: level / capacity <= pi/2 takes the short polynomial path (minimax since round 4), anything else the fp64-reduced general path.  Drive
    both through the env kernels (transition: sine; f_x: sine and cosine) with capacities of 1 so that the argument IS
    the state: physical levels, arguments beyond pi/2, negative and large ones, against numpy in fp64."""
    n = 4
    cfg = dict(problems.reservoir_config(n, seed=2, max_res_cap=1.0))
    env = Reservoir(**cfg)
    D, rain = env.downstream.astype(np.float64), env.rain_shape.astype(np.float64) * env.rain_scale
    rng = np.random.default_rng(0)
    levels = np.concatenate([rng.uniform(0.0, 1.5707, size=64), rng.uniform(-1.5707, 0.0, size=16),        # fast path
                             rng.uniform(1.5708, 10.0, size=64), rng.uniform(-300.0, 300.0, size=64),      # general path
                             rng.uniform(1e3, 1e5, size=32), [0.0, 1.5707963, 1.5707964, -1.5707964, 3.1415927, 1e6, -1e6, 12345.678]])
    levels = levels[: (len(levels) // n) * n].astype(np.float32).reshape(-1, n, 1)
    u = rng.uniform(size=levels.shape).astype(np.float32)
    lin = env.get_linear_transition(levels, u, batch=True)
    nxt = env.transition(levels, u, batch=True)
    for t in range(len(levels)):
        xt, ut = levels[t].astype(np.float64), u[t].astype(np.float64)
        ref = xt + rain - 0.5 * np.sin(xt) * xt - ut * xt + D.T @ (ut * xt)
        scale = np.abs(xt).max() + 1.0
        assert np.abs(_np(nxt[t]) - ref).max() <= 4e-7 * scale, (t, xt.ravel())           # |sin error| <= ~2 ulp of 1
        assert np.abs(_np(lin.f[t]) - ref).max() <= 4e-7 * scale
        fx = np.eye(n) - np.diag((0.5 * (np.cos(xt) * xt + np.sin(xt)))[:, 0]) - np.diag(ut[:, 0]) + D.T @ np.diag(ut[:, 0])
        assert np.abs(_np(lin.f_x[t]) - fx).max() <= 4e-7 * scale, (t, xt.ravel())


# ------------------------------------------------------------ test_env_navigation.py ---
@pytest.mark.parametrize("zones", [1, 2])
def test_navigation_closed_forms(zones):
    env = _navigation(zones, seed=zones)
    center, decay = np.asarray(env.deceleration["center"], float).reshape(zones, 2, 1), np.asarray(env.deceleration["decay"], float)
    goal = np.array([[8.0], [9.0]])
    rng = np.random.default_rng(zones)
    x = rng.normal(2.0, 1.0, size=(6, 2, 1)).astype(np.float32)
    u = rng.uniform(-1, 1, size=(6, 2, 1)).astype(np.float32)
    lin = env.get_linear_transition(x, u, batch=True)
    quad = env.get_quadratic_cost(x, u, batch=True)
    for t in range(len(x)):
        xt, ut = x[t].astype(np.float64), u[t].astype(np.float64)
        r = np.linalg.norm(xt - center, axis=(1, 2))
        lam_z = 2.0 / (1.0 + np.exp(-decay * r)) - 1.0                                         # :6-30
        lam = np.prod(lam_z)
        _close(lin.f[t], xt + lam * ut, "nav.f")                                               # :33-60
        _close(lin.f_u[t], lam * np.eye(2), "nav.f_u")
        h = 2.0 * decay * np.exp(-decay * r) / (1.0 + np.exp(-decay * r)) ** 2
        grad = sum(h[z] * (xt - center[z]) / r[z] * lam / lam_z[z] for z in range(zones))
        _close(lin.f_x[t], np.eye(2) + ut @ grad.T, "nav.f_x")                                  # :86-99
        _close(quad.l[t], ((xt - goal) ** 2).sum(), "nav.l")                                    # :102-130
        _close(quad.l_x[t], 2 * (xt - goal), "nav.l_x")
        _close(quad.l_xx[t], 2 * np.eye(2), "nav.l_xx")
        for h_ in ("l_u", "l_uu", "l_ux", "l_xu"):
            assert not _np(getattr(quad, h_)[t]).any()
    fin = env.get_quadratic_final_cost(x[0])
    _close(fin.l_x, 2 * (x[0].astype(np.float64) - goal), "nav.fl_x")                            # :150-176
    _close(fin.l_xx, 2 * np.eye(2), "nav.fl_xx")


# -------------------------------------------------------- test_env_lqr_navigation.py ---
def test_navigation_lqr_closed_forms():
    goal, beta = np.array([[5.5], [-9.0]]), 5.0
    env = NavigationLQR(goal, beta)
    x, u = _sample(env, 6, seed=1)
    lin = env.get_linear_transition(x, u, batch=True)
    quad = env.get_quadratic_cost(x, u, batch=True)
    I = np.eye(2)
    for t in range(len(x)):
        xt, ut = x[t].astype(np.float64), u[t].astype(np.float64)
        _close(lin.f[t], xt + ut, "navlqr.f")                                                   # :28-60
        assert np.array_equal(_np(lin.f_x[t]), I) and np.array_equal(_np(lin.f_u[t]), I)
        _close(quad.l[t], ((xt - goal) ** 2).sum() + beta * (ut ** 2).sum(), "navlqr.l")        # :63-100
        _close(quad.l_x[t], 2 * (xt - goal), "navlqr.l_x")
        _close(quad.l_u[t], 2 * beta * ut, "navlqr.l_u")
        _close(quad.l_xx[t], 2 * I, "navlqr.l_xx")
        _close(quad.l_uu[t], 2 * beta * I, "navlqr.l_uu")
        assert not _np(quad.l_ux[t]).any() and not _np(quad.l_xu[t]).any()
    fin = env.get_quadratic_final_cost(x[0])                                                    # :103-135
    _close(fin.l_x, 2 * (x[0].astype(np.float64) - goal), "navlqr.fl_x")
    _close(fin.l_xx, 2 * I, "navlqr.fl_xx")
