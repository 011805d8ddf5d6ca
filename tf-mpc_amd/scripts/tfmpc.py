#!/usr/bin/env python
"""Command line of the solver path -- same commands, arguments and options as the reference's
``scripts/tfmpc.py:26-215`` (``tfmpc lqr``, ``tfmpc navlin``, ``tfmpc ilqr [--online]``), on top of
``tfmpc.envs`` / ``tfmpc.launchers``.  Run as ``python tf-mpc_amd/scripts/tfmpc.py ...`` or
``python -m tfmpc ...`` (with ``tf-mpc_amd`` on ``sys.path``).

Differences, both forced by the platform: the ``--num-samples`` runs of ``tfmpc ilqr`` are one batch of
independent solves in ONE kernel launch on the GPU instead of a pool of ``--num-workers`` processes (the
option is accepted and ignored), and the results go to ``<logdir>/data.csv`` (one sample) or
``<logdir>/run<i>/data.csv`` without tuneconfig's trial directories.  ``--verbose`` / ``--debug`` set the
level of the ``tfmpc`` logger."""

import json
import logging
import os
import sys

import click
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _vector(text):
    return np.array(list(map(float, text.split())), dtype=np.float32)[:, np.newaxis]


def _verbosity(verbose=0, debug=False):
    level = logging.DEBUG if debug or verbose >= 2 else logging.INFO if verbose else logging.ERROR
    logging.getLogger("tfmpc").setLevel(level)


def _show(trajectory):
    print(repr(trajectory))
    print()
    print(str(trajectory))


@click.group()
def cli():
    pass


@cli.command()
@click.argument("initial-state")
@click.option("--action-size", "-a", type=click.IntRange(min=1), default=1, help="The number of action variables.")
@click.option("--horizon", "-hr", type=click.IntRange(min=1), default=10, help="The number of timesteps.")
@click.option("--debug", is_flag=True, help="Debug flag.")
@click.option("--verbose", "-v", is_flag=True, help="Verbosity flag.")
def lqr(initial_state, action_size, horizon, debug, verbose):
    """Generate and solve a randomly-created LQR problem.

    Args:

        initial_state: list of floats.
    """
    from tfmpc import envs

    _verbosity(int(verbose), debug)
    x0 = _vector(initial_state)
    solver = envs.make_lqr(x0.shape[0], action_size)
    _show(solver.solve(x0, horizon))


@cli.command()
@click.argument("initial-state")
@click.argument("goal")
@click.option("--beta", "-b", type=float, default=1.0, help="The weight of the action cost.")
@click.option("--horizon", "-hr", type=click.IntRange(min=1), default=10, help="The number of timesteps.")
@click.option("--debug", is_flag=True, help="Debug flag.")
@click.option("--verbose", "-v", is_flag=True, help="Verbosity flag.")
def navlin(initial_state, goal, beta, horizon, debug, verbose):
    """Generate and solve the linear navigation LQR problem.

    Args:

        initial_state: list of floats.

        goal: list of floats.
    """
    from tfmpc import envs

    _verbosity(int(verbose), debug)
    x0, g = _vector(initial_state), _vector(goal)
    if x0.shape != g.shape:
        raise click.BadParameter("initial state and goal must have the same size")
    solver = envs.make_lqr_linear_navigation(g, beta)
    _show(solver.solve(x0, horizon))


@cli.command()
@click.argument("env", type=click.Path(exists=True))
@click.option("--online", is_flag=True, help="Online mode flag.", show_default=True)
@click.option("--horizon", "-hr", type=click.IntRange(min=1), default=10, help="The number of timesteps.",
              show_default=True)
@click.option("--atol", type=click.FloatRange(min=0.0), default=5e-3, help="Absolute tolerance for convergence.",
              show_default=True)
@click.option("--max-iterations", "-miter", type=click.IntRange(min=1), default=100,
              help="Maximum number of iterations.", show_default=True)
@click.option("--logdir", type=click.Path(), default="/tmp/ilqr/", help="Directory used for logging results.",
              show_default=True)
@click.option("--num-samples", "-ns", type=click.IntRange(min=1), default=1,
              help="Number of runs (solved as one batch on the GPU).", show_default=True)
@click.option("--num-workers", "-nw", type=click.IntRange(min=1), default=1,
              help="Accepted for compatibility; the runs share one kernel launch.", show_default=True)
@click.option("--seed", type=int, default=None, help="Seed of the random initial actions and of the env noise "
              "(build addition).")
@click.option("--warm-start", is_flag=True, help="Online mode: start each re-solve from the shifted previous plan "
              "(build addition; the reference cold-starts).")
@click.option("--verbose", "-v", count=True,
              help="Verbosity level flag. -v also writes <logdir>/trace.log from the decision trace of the solve; its first line names the "
                   "kernel that recorded it (HVAC / Reservoir: a traced solve may run on another kernel family than an untraced one).")
def ilqr(env, online, horizon, atol, max_iterations, logdir, num_samples, num_workers, seed, warm_start, verbose):
    """Run iLQR for a given environment and horizon.

    Args:

        ENV: Path to the environment's config JSON file.
    """
    from tfmpc import agents, envs, runners
    from tfmpc.solvers import ilqr as ilqr_solver

    _verbosity(verbose)
    with open(env, "r") as file:
        env_config = json.load(file)
    model = envs.make_env(env_config)
    x0 = np.asarray(env_config["initial_state"], dtype=np.float32).reshape(-1, 1)
    if num_samples > 1:
        x0 = np.broadcast_to(x0, (num_samples,) + x0.shape).copy()
    os.makedirs(logdir, exist_ok=True)
    solver = ilqr_solver.iLQR(model, atol=atol, max_iterations=max_iterations)

    if online:
        controller = agents.MPC(solver, horizon, warm_start=warm_start, seed=seed)
        if seed is not None:
            model.seed(seed)
        with runners.Runner(model, controller)(x0, horizon) as r:
            trajectory = r.run()
    else:
        # -v: what the reference logs while it solves (ilqr.py:41-43 -> <logdir>/trace.log; :279 progress postfix), from the
        # decision trace of the fused solve
        trajectory, _ = solver.solve(x0, horizon, seed=seed, trace=verbose >= 1, show_progress=num_samples <= 1)
        if verbose >= 1:
            with open(os.path.join(logdir, "trace.log"), "w") as file:
                # (the trace is recorded by the kernel that solves; for HVAC / Reservoir that can be another kernel family -- other
                # rounding, possibly other line-search decisions -- than the run without -v takes: say which one this was)
                file.write(f"[KERNEL] {getattr(solver, 'last_kernel', 'host-driven loop (generic env)')}\n")
                for b, records in enumerate(solver.last_trace):
                    if len(solver.last_trace) > 1:
                        file.write(f"[SAMPLE] {b}\n")
                    file.write("\n".join(ilqr_solver.trace_log_lines(records)) + "\n")

    runs = [trajectory] if not trajectory.batched else [trajectory.instance(b) for b in range(num_samples)]
    for i, run in enumerate(runs):
        run.save(os.path.join(logdir, "data.csv") if len(runs) == 1 else os.path.join(logdir, f"run{i}", "data.csv"))
        print(repr(run))
        print(str(run))


if __name__ == "__main__":
    cli()
