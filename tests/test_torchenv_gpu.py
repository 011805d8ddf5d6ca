"""Generic differentiable envs given as torch functions (the reference's
"any TensorFlow-differentiable env", tfmpc/envs/diffenv.py:13-101): torch.func derivatives +
the HIP Riccati backward pass + batched torch rollouts must reproduce what the built-in
device-resident envs produce for the same model."""

import numpy as np
import pytest
import torch

import problems
from tfmpc.envs.lqr.navigation import NavigationLQR
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.torchenv import TorchEnv
from tfmpc.solvers.ilqr import iLQR

pytestmark = pytest.mark.gpu


def _navlqr_torch(goal, beta, low=None, high=None):
    g = torch.as_tensor(goal, dtype=torch.float32, device="cuda").reshape(-1)
    return TorchEnv(lambda x, u: x + u,
                    lambda x, u: ((x - g) ** 2).sum() + beta * (u ** 2).sum(),
                    lambda x: ((x - g) ** 2).sum(), 2, 2, low, high)


def _navigation_torch(cfg):
    g = torch.as_tensor(cfg["goal"], dtype=torch.float32, device="cuda").reshape(-1)
    centers = torch.as_tensor(cfg["deceleration"]["center"], dtype=torch.float32, device="cuda").reshape(-1, 2)
    decay = torch.as_tensor(cfg["deceleration"]["decay"], dtype=torch.float32, device="cuda")

    def transition(x, u):
        dist = torch.linalg.norm(x[None, :] - centers, dim=-1)
        lam = torch.prod(2.0 / (1.0 + torch.exp(-decay * dist)) - 1.0)
        return x + lam * u

    cost = lambda x, u: ((x - g) ** 2).sum()
    return TorchEnv(transition, cost, lambda x: ((x - g) ** 2).sum(), 2, 2,
                    np.asarray(cfg["low"]).reshape(2, 1), np.asarray(cfg["high"]).reshape(2, 1))


def test_torch_func_models_match_the_device_closed_forms():
    cfg = problems.NAV_CONFIG
    builtin, generic = Navigation.load(cfg), _navigation_torch(cfg)
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 10, size=(9, 2, 1)).astype(np.float32)
    u = rng.uniform(-1, 1, size=(9, 2, 1)).astype(np.float32)
    a, b = builtin.get_linear_transition(x, u), generic.get_linear_transition(x, u)
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
    a, b = builtin.get_quadratic_cost(x, u), generic.get_quadratic_cost(x, u)
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
    a, b = builtin.get_quadratic_final_cost(x[0]), generic.get_quadratic_final_cost(x[0])
    for p, q in zip(a, b):
        assert torch.allclose(p, q.reshape(p.shape), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bounds", [None, (-1.0, 1.0)])
def test_generic_env_solve_equals_builtin_on_linear_navigation(bounds):
    low, high = bounds if bounds else (None, None)
    goal, beta, B, T = [[5.5], [-9.0]], 5.0, 12, 10
    sb, sg = iLQR(NavigationLQR(goal, beta, low, high)), iLQR(_navlqr_torch(goal, beta, low, high))
    x0 = np.random.default_rng(2).normal(size=(B, 2, 1)).astype(np.float32)
    u0 = sb.random_actions(T, B, seed=4)
    tb, ib = sb.solve(x0, T, u_init=u0)
    tg, ig = sg.solve(x0, T, u_init=u0)
    assert np.array_equal(ib, ig)
    assert np.abs(tb.states - tg.states).max() <= 1e-3 * np.abs(tb.states).max()
    assert np.abs(tb.costs - tg.costs).max() <= 1e-3 * np.abs(tb.costs).max()
    # the piecewise API works on generic envs too (reference tests/test_ilqr.py:48-109)
    xs, us, cs = sg.start(x0[0], T, u_init=u0[0])
    models = sg.derivatives(xs, us)
    K, k, J, dV1, dV2 = sg.backward(T, us, *models)
    st, ac, co, Jn, res = sg.forward(xs, us, K, k)
    assert st.shape == xs.shape and ac.shape == us.shape and co.shape == (T + 1,)
    assert torch.allclose(st[1], sg.env.transition(st[0], ac[0]))


def test_generic_env_solve_tracks_builtin_on_nonlinear_navigation():
    cfg = problems.NAV_CONFIG
    sb, sg = iLQR(Navigation.load(cfg)), iLQR(_navigation_torch(cfg))
    B, T = 24, 20
    x0 = np.random.default_rng(3).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = sb.random_actions(T, B, seed=6)
    tb, ib = sb.solve(x0, T, u_init=u0)
    tg, ig = sg.solve(x0, T, u_init=u0)
    rel = np.abs(tb.total_cost - tg.total_cost) / np.abs(tb.total_cost)
    assert np.median(rel) <= 1e-4 and np.quantile(rel, 0.9) <= 2e-2       # a few flipped line searches
    assert np.mean(ib == ig) >= 0.7
    assert np.abs(tg.actions).max() <= 1.0
