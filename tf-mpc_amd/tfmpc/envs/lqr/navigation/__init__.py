"""Linear-quadratic navigation -- drop-in for the reference's
``tfmpc/envs/lqr/navigation/__init__.py:8-68``: ``x' = x + u``,
``cost = |x - g|^2 + beta |u|^2``, optional scalar box bounds on the action."""

import numpy as np

from tfmpc import _hip
from tfmpc.envs.diffenv import Box, DiffEnv


class NavigationLQR(DiffEnv):
    kind = _hip.ENV_NAVLQR

    def __init__(self, goal, beta, low=None, high=None):
        goal = np.asarray(goal.cpu() if hasattr(goal, "cpu") else goal, dtype=np.float32)
        if goal.shape[-1] != 1:
            goal = goal[..., None]
        self.goal = goal                                  # [n,1] or [B,n,1] (per-instance goals)
        self.beta = float(beta)
        low = -np.inf if low is None else low
        high = np.inf if high is None else high
        n = goal.shape[-2]
        self.obs_space = Box(-np.inf, np.inf, (n, 1))
        self.action_space = Box(low, high, (n, 1))
        self.scalars = (self.beta,)

    @property
    def state_size(self):
        return self.goal.shape[-2]

    @property
    def action_size(self):
        return self.state_size

    def _params(self):
        return [(self.goal[..., 0], 1)]

    @classmethod
    def load(cls, config):
        goal = np.asarray(config["goal"], dtype=np.float32).reshape(len(config["goal"]), 1)
        return cls(goal, config["beta"], config.get("low"), config.get("high"))

    def __repr__(self):
        bounds = ""
        if self.action_space.is_bounded():
            bounds = f", bounds=[{self.action_space.low.squeeze().tolist()}, {self.action_space.high.squeeze().tolist()}]"
        return f"NavigationLQR(goal={self.goal.squeeze().tolist()}, beta={self.beta}{bounds})"
