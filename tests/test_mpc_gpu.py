"""Online (receding-horizon) MPC loop -- the first "next" row after the solver path
(reference: tfmpc/agents/mpc.py:4-15, tfmpc/runners/__init__.py:8-49,
tfmpc/envs/gymenv.py:4-41, tfmpc/launchers/__init__.py:31-51) -- batched over episodes."""

import json

import numpy as np
import pytest
import torch

import problems
from tfmpc import agents, launchers, runners
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

pytestmark = pytest.mark.gpu


def _episode(env, x0, T, **mpc_kwargs):
    agent = agents.MPC(iLQR(env), T, **mpc_kwargs)
    runner = runners.Runner(env, agent)
    with runner(x0, T) as r:
        traj = r.run()
    return traj, agent


def test_certainty_equivalent_mpc_reproduces_the_open_loop_plan():
    """With deterministic stepping and warm starts, re-planning from the planned state follows
    the original plan (principle of optimality) and each re-solve converges almost at once."""
    env = Navigation.load(problems.NAV_CONFIG)
    env.stochastic = False
    B, T = 6, 15
    x0 = np.random.default_rng(0).uniform(0, 6, size=(B, 2, 1)).astype(np.float32)
    solver = iLQR(env)
    u0 = solver.random_actions(T, B, seed=1)
    plan, _ = solver.solve(x0, T, u_init=u0)
    agent = agents.MPC(solver, T, warm_start=True)
    agent._plan = torch.cat([u0[:, :1], u0], dim=1)       # so the first solve starts from the same actions
    runner = runners.Runner(env, agent)
    env.setup(x0, T)
    state, done, t = env.reset(), False, 0
    states, costs = [state], []
    while not done:
        action = agent(state, t)
        state, cost, done, _ = env.step(action)
        t = env._t
        states.append(state)
        costs.append(cost)
    total = torch.stack(costs, dim=1).sum(dim=1) + env.final_cost(state, batch=True)
    assert np.all(total.cpu().numpy() <= plan.total_cost * 1.02 + 1e-3)
    assert np.mean([np.mean(it) for it in agent.iterations[1:]]) <= 3.0      # warm-started re-solves are cheap


def test_stochastic_batched_episode_shapes_seeds_and_consistency():
    env = Navigation.load(problems.NAV_CONFIG)
    B, T = 16, 10
    x0 = np.random.default_rng(1).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    runs = []
    for _ in range(2):
        env.seed(123)
        traj, agent = _episode(env, x0, T, warm_start=True, seed=7)
        runs.append(traj)
    a, b = runs
    assert a.states.shape == (B, T + 1, 2) and a.actions.shape == (B, T, 2) and a.costs.shape == (B, T + 1)
    assert np.array_equal(a.states, b.states) and np.array_equal(a.costs, b.costs)      # reproducible per seed
    assert np.abs(a.actions).max() <= 1.0
    # every step is the env's transition plus truncated-normal noise within two sigma (navigation/__init__.py:45)
    st = torch.as_tensor(a.states, device="cuda").unsqueeze(-1)
    ac = torch.as_tensor(a.actions, device="cuda").unsqueeze(-1)
    for t in (0, 4, 9):
        det = env.transition(st[:, t], ac[:, t], batch=True)
        noise = (st[:, t + 1] - det).abs()
        assert float(noise.max()) <= 0.4 + 1e-5 and float(noise.max()) > 0.0
        assert torch.allclose(env.cost(st[:, t], ac[:, t], batch=True).cpu(), torch.as_tensor(a.costs[:, t]), rtol=1e-6)
    # most noisy episodes still end near the goal
    assert np.median(a.costs[:, -1]) < 10.0


def test_single_episode_api_and_reservoir_noise():
    env = Reservoir.load(dict(problems.RES4_CONFIG))
    env.seed(5)
    T = 6
    traj, agent = _episode(env, np.asarray(problems.RES4_X0, dtype=np.float32), T, seed=3)
    assert traj.states.shape == (T + 1, 4) and traj.actions.shape == (T, 4) and traj.costs.shape == (T + 1,)
    assert len(agent.iterations) == T and all(isinstance(i, int) for i in agent.iterations)
    assert np.all(traj.actions >= 0.0) and np.all(traj.actions <= 1.0)
    assert np.isfinite(traj.states).all()


def test_online_launcher_writes_reference_csv(tmp_path):
    cfg_path = tmp_path / "nav.config.json"
    cfg_path.write_text(json.dumps({"module": "navigation", "cls_name": "Navigation", "config": problems.NAV_CONFIG,
                                    "initial_state": [[0.0], [0.0]]}))
    env, traj = launchers.online_ilqr_run({"env": str(cfg_path), "horizon": 8, "logdir": str(tmp_path / "run"),
                                           "atol": 5e-3, "warm_start": True})
    import pandas as pd
    df = pd.read_csv(tmp_path / "run" / "data.csv")
    assert list(df.columns) == ["Timestep", "x[1]", "x[2]", "u[1]", "u[2]", "costs"] and len(df) == 8
    env2, traj2 = launchers.ilqr_run({"env": str(cfg_path), "horizon": 8, "logdir": str(tmp_path / "run2")})
    assert len(traj2) == 8 and (tmp_path / "run2" / "data.csv").exists()
