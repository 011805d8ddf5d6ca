"""Nonlinear 2-D navigation with deceleration zones -- drop-in for the reference's
``tfmpc/envs/navigation/__init__.py:9-98`` (deterministic ``cec=True`` dynamics):
``x' = x + lambda(x) u``, ``lambda = prod_z (2 / (1 + exp(-decay_z |x - c_z|)) - 1)``,
``cost = |x - g|^2``, box-bounded actions."""

import numpy as np

from tfmpc import _hip
from tfmpc.envs.diffenv import Box, DiffEnv
from tfmpc.envs.gymenv import GymEnv


def _np(a):
    return np.asarray(a.cpu() if hasattr(a, "cpu") else a, dtype=np.float32)


class Navigation(DiffEnv, GymEnv):
    kind = _hip.ENV_NAVIGATION

    NOISE_STDDEV = 0.2      # tf.random.truncated_normal(stddev=0.2), navigation/__init__.py:45

    def __init__(self, goal, deceleration, low, high):
        self._gym_init()
        goal = _np(goal)
        if goal.shape[-1] != 1:
            goal = goal[..., None]
        self.goal = goal                                               # [n,1] or [B,n,1]
        n = goal.shape[-2]
        self.deceleration = {"center": _np(deceleration["center"]).reshape(-1, n, 1),
                             "decay": _np(deceleration["decay"]).reshape(-1)}
        self.n_zones = self.deceleration["center"].shape[0]
        self.obs_space = Box(-np.inf, np.inf, (n, 1))
        # bounds normalised to the action column [m,1] (the 1-D form of the reference's
        # tests/conftest.py:149-151 would broadcast clip to [2,2], quirk Q10)
        self.action_space = Box(_np(low).reshape(n, 1), _np(high).reshape(n, 1), (n, 1))

    @property
    def state_size(self):
        return self.goal.shape[-2]

    @property
    def action_size(self):
        return self.state_size

    def _params(self):
        return [(self.goal[..., 0], 1), (self.deceleration["center"][..., 0], 2), (self.deceleration["decay"], 1)]

    def _noise(self, state):
        import torch
        eps = torch.empty_like(state)
        # TF's truncated normal re-draws samples beyond two standard deviations
        torch.nn.init.trunc_normal_(eps, mean=0.0, std=self.NOISE_STDDEV, a=-2 * self.NOISE_STDDEV,
                                    b=2 * self.NOISE_STDDEV, generator=self._generator)
        return eps

    @classmethod
    def load(cls, config):
        return cls(np.asarray(config["goal"], dtype=np.float32),
                   {k: np.asarray(v, dtype=np.float32) for k, v in config["deceleration"].items()},
                   config["low"], config["high"])

    def __repr__(self):
        return (f"Navigation(goal={self.goal.squeeze().tolist()}, zones={self.n_zones}, "
                f"bounds=[{self.action_space.low.squeeze().tolist()}, {self.action_space.high.squeeze().tolist()}])")
