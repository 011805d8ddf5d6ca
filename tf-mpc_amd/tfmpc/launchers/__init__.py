"""Glue from a config dict to a solved trajectory -- drop-in for the reference's
``tfmpc/launchers/__init__.py:12-51`` (``ilqr_run``, ``online_ilqr_run``): read the env JSON,
build the solver, run, save ``<logdir>/data.csv``."""

import json
import os

import numpy as np

from tfmpc import agents, envs, runners
from tfmpc.solvers import ilqr


def _load(config):
    config = dict(config)
    env_config = config.pop("env")
    if isinstance(env_config, str):
        with open(env_config, "r") as file:
            env_config = json.load(file)
    env = envs.make_env(env_config)
    x0 = np.asarray(env_config["initial_state"], dtype=np.float32)
    T = int(config.pop("horizon"))
    return env, x0, T, config


def ilqr_run(config):
    env, x0, T, config = _load(config)
    solver = ilqr.iLQR(env, **config)
    trajectory, iterations = solver.solve(x0, T)
    if "logdir" in config:
        trajectory.save(os.path.join(config["logdir"], "data.csv"))
    return env, trajectory


def online_ilqr_run(config):
    env, x0, T, config = _load(config)
    warm_start = config.pop("warm_start", False)
    solver = ilqr.iLQR(env, **config)
    controller = agents.MPC(solver, T, warm_start=warm_start)
    runner = runners.Runner(env, controller)
    with runner(x0, T) as r:
        trajectory = r.run()
        if "logdir" in config:
            trajectory.save(os.path.join(config["logdir"], "data.csv"))
    return env, trajectory
