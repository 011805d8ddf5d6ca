"""n-room HVAC thermal model -- drop-in for the reference's
``tfmpc/envs/hvac/__init__.py:8-195``: bilinear dynamics in (temperature, air flow),
piecewise-linear cost, actions in [0, 1]."""

import numpy as np

from tfmpc import _hip
from tfmpc.envs.diffenv import Box, DiffEnv


def _np(a, dtype=np.float32):
    return np.asarray(a.cpu() if hasattr(a, "cpu") else a, dtype=dtype)


class HVAC(DiffEnv):
    kind = _hip.ENV_HVAC
    CAP_AIR, COST_AIR, TEMP_AIR, TIME_DELTA = 1.006, 1.0, 40.0, 1.0
    PENALTY, SET_POINT_PENALTY = 20_000.0, 10.0

    def __init__(self, temp_outside, temp_hall, temp_lower_bound, temp_upper_bound, R_outside, R_hall, R_wall,
                 capacity, air_max, adj, adj_outside, adj_hall):
        col = lambda a: _np(a).reshape(-1, 1)
        self.temp_outside, self.temp_hall = col(temp_outside), col(temp_hall)
        self.temp_lower_bound, self.temp_upper_bound = col(temp_lower_bound), col(temp_upper_bound)
        self.R_outside, self.R_hall, self.R_wall = col(R_outside), col(R_hall), _np(R_wall)
        self.capacity, self.air_max = col(capacity), col(air_max)
        self.adj = _np(adj, bool)
        self.adj_outside = _np(adj_outside, bool).reshape(-1, 1)
        self.adj_hall = _np(adj_hall, bool).reshape(-1, 1)
        n = self.state_size
        self.obs_space = Box(-np.inf, np.inf, (n, 1))
        self.action_space = Box(0.0, 1.0, (n, 1))

    @property
    def state_size(self):
        return len(self.temp_lower_bound)

    @property
    def action_size(self):
        return self.state_size

    def _params(self):
        # conductances in fp32 like the reference's tensors (hvac/__init__.py:131-149)
        sym = np.logical_or(self.adj, self.adj.T).astype(np.float32)
        G = (sym / self.R_wall).astype(np.float32)
        k_out = (self.adj_outside.astype(np.float32) / self.R_outside).astype(np.float32)
        k_hall = (self.adj_hall.astype(np.float32) / self.R_hall).astype(np.float32)
        vec = lambda a: (a[:, 0], 1)
        return [vec(self.temp_outside), vec(self.temp_hall), vec(self.temp_lower_bound), vec(self.temp_upper_bound),
                vec(k_out), vec(k_hall), vec(self.capacity), vec(self.air_max), (G, 2)]

    @classmethod
    def load(cls, config):
        return cls(**{k: np.asarray(v, dtype=bool if k.startswith("adj") else np.float32) for k, v in config.items()})

    def __repr__(self):
        return f"HVAC({self.state_size})"

    def __str__(self):
        """Parameter listing with the reference's fields (``hvac/__init__.py:154-186``)."""
        row = lambda a, fmt=".3f": "[" + ", ".join(format(float(v), fmt) for v in np.ravel(a)) + "]"
        bounds = ", ".join(f"[{lo:.3f}, {hi:.3f}]" for lo, hi in zip(np.ravel(self.temp_lower_bound), np.ravel(self.temp_upper_bound)))
        R = "\n".join(f"[outside={o:.3f}, hall={h:.3f}]" for o, h in zip(np.ravel(self.R_outside), np.ravel(self.R_hall)))
        fields = [f"temp_bounds=[{bounds}]", f"R=\n{R}", f"R_wall=\n{self.R_wall}", f"capacity={row(self.capacity)}",
                  f"air_max={row(self.air_max)}", f"adj=\n{self.adj}", f"adj_outside={np.ravel(self.adj_outside).tolist()}",
                  f"adj_hall={np.ravel(self.adj_hall).tolist()}", f"temp_outside={row(self.temp_outside)}",
                  f"temp_hall={row(self.temp_hall)}"]
        return "HVAC(\n" + ",\n".join(fields) + "\n)"
