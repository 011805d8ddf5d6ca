"""Problem factories -- drop-in for the reference's ``tfmpc/envs/__init__.py``
(``make_lqr`` :9-18, ``make_lqr_linear_navigation`` :21-30, ``make_env`` :33-37)."""

import importlib

import numpy as np

from tfmpc.solvers.lqr import LQR


def make_lqr(state_size, action_size, batch_size=None, device=None):
    """Random LQR problem(s).  Draw order from the GLOBAL numpy RNG is the
    reference's: F, f, ``make_spd_matrix(n+m)``, c (per instance when batched)."""
    from sklearn.datasets import make_spd_matrix

    n_dim = state_size + action_size

    def one():
        F = np.random.normal(size=(state_size, n_dim))
        f = np.random.normal(size=(state_size, 1))
        C = make_spd_matrix(n_dim)
        c = np.random.normal(size=(n_dim, 1))
        return F, f, C, c

    if batch_size is None:
        return LQR(*one(), device=device)
    return LQR(*(np.stack(a) for a in zip(*[one() for _ in range(batch_size)])), device=device)


def make_lqr_linear_navigation(goal, beta, device=None):
    """``x' = x + u``, cost ``|x-g|^2 + beta |u|^2`` with the constant dropped.
    ``goal`` is ``[n,1]`` or ``[B,n,1]``: with a batch only ``c`` varies, ``F`` and
    ``C`` are shared by all instances.  ``F = [I I]`` for any n (the reference's
    ``[I]*action_size`` is only right for n = 2, quirk Q7)."""
    goal = np.asarray(goal, dtype=np.float32)
    if goal.shape[-1] != 1:
        goal = goal[..., None]
    n = goal.shape[-2]
    F = np.concatenate([np.identity(n), np.identity(n)], axis=1).astype("f")
    f = np.zeros((n, 1)).astype("f")
    C = np.diag([2.0] * n + [2.0 * beta] * n).astype("f")
    c = np.concatenate([-2.0 * goal, np.zeros_like(goal)], axis=-2).astype("f")
    return LQR(F, f, C, c, device=device)


def make_env(config):
    """``{"module", "cls_name", "config"}`` -> env instance.  Also accepts the module
    name ``navigation_lqr`` that the reference's ``navlin.config.json:2`` uses although
    the package is ``tfmpc.envs.lqr.navigation`` (quirk Q8)."""
    module = config["module"]
    module = {"navigation_lqr": "lqr.navigation"}.get(module, module)
    module = importlib.import_module(f"tfmpc.envs.{module}")
    return getattr(module, config["cls_name"]).load(config["config"])
