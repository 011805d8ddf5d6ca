"""A whole MPC episode captured as one hipGraph (``runners.Runner.capture``) reproduces the eager
loop BIT FOR BIT -- same launches, same inputs, no host in between -- for the initial states and noise draws it was
captured with and for new ones copied into its static buffers.  (The eager loop itself is checked against the oracle's
restatement of the reference's loop in tests/test_mpc_oracle_gpu.py.)"""

import numpy as np
import pytest
import torch

import problems
from tfmpc import agents, runners
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

pytestmark = pytest.mark.gpu


def _inputs(kind, B, T, seed):
    rng = np.random.default_rng(seed)
    if kind == "navigation":
        x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
        noise = [np.clip(rng.normal(0.0, 0.2, size=(B, 2, 1)), -0.4, 0.4).astype(np.float32) for _ in range(T)]
    else:
        x0 = (np.array(problems.RES4_X0)[None] * rng.uniform(0.8, 1.2, size=(B, 4, 1))).astype(np.float32)
        shape, scale = np.array(problems.RES4_CONFIG["rain_shape"]), np.array(problems.RES4_CONFIG["rain_scale"])
        noise = [rng.gamma(shape, scale, size=(B, 4, 1)).astype(np.float32) for _ in range(T)]
    return x0, noise


def _env(kind):
    return Navigation.load(problems.NAV_CONFIG) if kind == "navigation" else Reservoir.load(dict(problems.RES4_CONFIG))


def _eager(kind, T, warm, x0, noise):
    env = _env(kind)
    env.inject_noise(noise)
    agent = agents.MPC(iLQR(env), T, warm_start=warm, seed=11)
    with runners.Runner(env, agent)(x0, T) as r:
        traj = r.run()
    return traj, np.stack([np.asarray(i).reshape(-1) for i in agent.iterations])


@pytest.mark.parametrize("kind,B,T,warm", [("navigation", 1, 8, True), ("navigation", 37, 6, False), ("reservoir", 5, 6, True)])
def test_captured_episode_equals_the_eager_loop(kind, B, T, warm):
    x0a, na = _inputs(kind, B, T, seed=1)
    x0b, nb = _inputs(kind, B, T, seed=2)
    env = _env(kind)
    agent = agents.MPC(iLQR(env), T, warm_start=warm, seed=11)
    episode = runners.Runner(env, agent).capture(x0a, T, na)
    for x0, noise in ((x0a, na), (x0b, nb), (x0a, na)):
        traj, its = episode(x0, noise)
        ref, ref_its = _eager(kind, T, warm, x0, noise)
        assert np.array_equal(traj.states, ref.states) and np.array_equal(traj.actions, ref.actions)
        assert np.array_equal(traj.costs, ref.costs) and np.array_equal(its, ref_its)


def test_capture_needs_one_draw_per_step():
    env = _env("navigation")
    x0, noise = _inputs("navigation", 2, 4, seed=3)
    with pytest.raises(ValueError):
        runners.Runner(env, agents.MPC(iLQR(env), 4, seed=1)).capture(x0, 4, noise[:3])
