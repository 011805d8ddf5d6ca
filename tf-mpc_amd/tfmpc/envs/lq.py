"""A time-invariant LQ problem ``(F, f, C, c)`` (``tfmpc/solvers/lqr.py:36-57`` of the
reference) presented through the DiffEnv protocol so that iLQR can be driven on shapes
with ``action_size != state_size`` -- the BASELINE.json headline shape n=16, m=8.  The
reference's own envs all have m == n; this env is the build's addition."""

import numpy as np

from tfmpc import _hip
from tfmpc.envs.diffenv import Box, DiffEnv


def _np(a):
    return np.asarray(a.cpu() if hasattr(a, "cpu") else a, dtype=np.float32)


class LQEnv(DiffEnv):
    kind = _hip.ENV_LQ

    def __init__(self, F, f, C, c, low=None, high=None):
        self.F, self.C = _np(F), _np(C)                       # [n,d] / [d,d], optionally with a leading B
        n, d = self.F.shape[-2:]
        self.f = self._vec(f, n)                               # [n] or [B,n]
        self.c = self._vec(c, d)
        self._n, self._m = n, d - n
        low = -np.inf if low is None else low
        high = np.inf if high is None else high
        self.obs_space = Box(-np.inf, np.inf, (n, 1))
        self.action_space = Box(low, high, (self._m, 1))

    @staticmethod
    def _vec(a, size):
        a = _np(a)
        if a.ndim >= 2 and a.shape[-1] == 1 and a.shape[-2] == size:
            a = a[..., 0]
        if a.shape[-1] != size or a.ndim not in (1, 2):
            raise ValueError(f"expected [{size}] or [B,{size}], got {a.shape}")
        return a

    @classmethod
    def from_lqr(cls, lqr, low=None, high=None):
        return cls(lqr.F, lqr.f[..., 0], lqr.C, lqr.c[..., 0], low, high)

    @property
    def state_size(self):
        return self._n

    @property
    def action_size(self):
        return self._m

    def _params(self):
        return [(self.F, 2), (self.f, 1), (self.C, 2), (self.c, 1)]
