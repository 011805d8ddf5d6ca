"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's differentiable
environments and of ``tfmpc/envs/diffenv.py``.

The reference obtains every Jacobian/Hessian by TensorFlow autodiff
(``diffenv.py:13-101``).  This restatement writes the env equations with torch
CPU ops and differentiates them with ``torch.autograd`` in the same way, so the
analytic device code in the product is checked against an INDEPENDENT autodiff
result, not against a second hand derivation.  Subgradient conventions match
TensorFlow's: ``d|y|/dy = sign(y)`` and ``d max(0, y)/dy = [y > 0]``
(``torch.relu``), see SURVEY.md Appendix A.3.

All states/actions are column vectors ``[n, 1]`` / ``[m, 1]`` (numpy in, numpy
out); time-batched inputs are ``[T, n, 1]``.
"""

from collections import namedtuple

import numpy as np
import torch

TransitionApprox = namedtuple("TransitionApprox", "f f_x f_u")              # diffenv.py:6
CostApprox = namedtuple("CostApprox", "l l_x l_u l_xx l_uu l_ux l_xu")      # diffenv.py:7
FinalCostApprox = namedtuple("FinalCostApprox", "l l_x l_xx")               # diffenv.py:8


class Box:
    """The two attributes and the one method of ``gym.spaces.Box`` that the hot
    path touches (``ilqr.py:47,51,136``)."""

    def __init__(self, low, high, shape):
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), shape).copy()

    def is_bounded(self):
        return bool(np.all(np.isfinite(self.low)) and np.all(np.isfinite(self.high)))


class OracleEnv:
    """Autodiff hooks of ``diffenv.py`` over torch-CPU restatements of
    ``transition`` / ``cost`` / ``final_cost``."""

    def __init__(self, dtype=np.float64):
        self.dtype = np.dtype(dtype).type
        self.tdtype = torch.float64 if self.dtype == np.float64 else torch.float32

    # -- to be provided by each env, torch [n,1],[m,1] -> torch --------------
    def _transition(self, x, u):
        raise NotImplementedError

    def _cost(self, x, u):
        raise NotImplementedError

    def _final_cost(self, x):
        raise NotImplementedError

    def _t(self, a):
        return torch.as_tensor(np.asarray(a, dtype=self.dtype), dtype=self.tdtype)

    # -- numpy API (single step) ---------------------------------------------
    def transition(self, state, action):
        with torch.no_grad():
            return self._transition(self._t(state), self._t(action)).numpy()

    def cost(self, state, action):
        with torch.no_grad():
            return self._cost(self._t(state), self._t(action)).numpy().reshape(())

    def final_cost(self, state):
        with torch.no_grad():
            return self._final_cost(self._t(state)).numpy().reshape(())

    # -- diffenv.py:13-32 -------------------------------------------------------
    # The reference differentiates the whole time axis at once (GradientTape.batch_jacobian, diffenv.py:21-22); so does
    # the restatement: torch.func's jacrev / hessian of the single-step functions, vmapped over the T steps (the same
    # reverse-mode derivatives as a per-step loop of torch.autograd.functional.jacobian, which `_loop=True` still runs --
    # tests/test_oracle_pinning.py holds the two against each other -- at a fiftieth of the time).
    def get_linear_transition(self, states, actions, _loop=False):
        n, m = self.state_size, self.action_size
        if not _loop and len(states) > 0:
            X, U = self._t(np.stack(states)), self._t(np.stack(actions))
            with torch.no_grad():
                f = torch.func.vmap(self._transition)(X, U)
            jx, ju = torch.func.vmap(torch.func.jacrev(self._transition, argnums=(0, 1)))(X, U)
            T = X.shape[0]
            return TransitionApprox(f.numpy(), jx.reshape(T, n, n).numpy(), ju.reshape(T, n, m).numpy())
        fs, fxs, fus = [], [], []
        for x, u in zip(states, actions):
            x, u = self._t(x), self._t(u)
            f = self._transition(x, u)
            jx, ju = torch.autograd.functional.jacobian(self._transition, (x, u))
            fs.append(f.detach().numpy())
            fxs.append(jx.reshape(n, n).numpy())
            fus.append(ju.reshape(n, m).numpy())
        return TransitionApprox(np.stack(fs), np.stack(fxs), np.stack(fus))

    # -- diffenv.py:34-83 -------------------------------------------------------
    def get_quadratic_cost(self, states, actions, _loop=False):
        n, m = self.state_size, self.action_size

        def scalar_cost(x, u):
            return self._cost(x, u).reshape(())

        if not _loop and len(states) > 0:
            X, U = self._t(np.stack(states)), self._t(np.stack(actions))
            T = X.shape[0]
            with torch.no_grad():
                l = torch.func.vmap(scalar_cost)(X, U)
            gx, gu = torch.func.vmap(torch.func.jacrev(scalar_cost, argnums=(0, 1)))(X, U)
            (hxx, hxu), (hux, huu) = torch.func.vmap(torch.func.jacfwd(torch.func.jacrev(scalar_cost, argnums=(0, 1)), argnums=(0, 1)))(X, U)
            return CostApprox(l.numpy(), gx.reshape(T, n, 1).numpy(), gu.reshape(T, m, 1).numpy(),
                              hxx.reshape(T, n, n).numpy(), huu.reshape(T, m, m).numpy(),
                              hux.reshape(T, m, n).numpy(), hxu.reshape(T, n, m).numpy())
        out = [[] for _ in range(7)]
        for x, u in zip(states, actions):
            x, u = self._t(x), self._t(u)
            l = scalar_cost(x, u)
            gx, gu = torch.autograd.functional.jacobian(scalar_cost, (x, u))
            (hxx, hxu), (hux, huu) = torch.autograd.functional.hessian(scalar_cost, (x, u))
            vals = (l.detach().numpy(), gx.reshape(n, 1).numpy(), gu.reshape(m, 1).numpy(),
                    hxx.reshape(n, n).numpy(), huu.reshape(m, m).numpy(),
                    hux.reshape(m, n).numpy(), hxu.reshape(n, m).numpy())
            for o, v in zip(out, vals):
                o.append(v)
        return CostApprox(*[np.stack(o) for o in out])

    # -- diffenv.py:85-101 ------------------------------------------------------
    def get_quadratic_final_cost(self, state):
        n = self.state_size
        x = self._t(state)

        def scalar_cost(x):
            return self._final_cost(x).reshape(())

        l = scalar_cost(x)
        gx = torch.autograd.functional.jacobian(scalar_cost, x)
        hxx = torch.autograd.functional.hessian(scalar_cost, x)
        return FinalCostApprox(l.detach().numpy(), gx.reshape(n, 1).numpy(), hxx.reshape(n, n).numpy())


class NavigationLQR(OracleEnv):
    """``tfmpc/envs/lqr/navigation/__init__.py:8-47``"""

    def __init__(self, goal, beta, low=None, high=None, dtype=np.float64):
        super().__init__(dtype)
        self.goal = np.asarray(goal, dtype=self.dtype).reshape(-1, 1)
        self.beta = self.dtype(beta)
        low = -np.inf if low is None else low                         # :14-17
        high = np.inf if high is None else high
        self.action_space = Box(low, high, self.goal.shape)           # :20
        self.state_size = self.action_size = self.goal.shape[0]

    def _transition(self, x, u):
        return x + u                                                  # :32

    def _cost(self, x, u):
        g = self._t(self.goal)
        return torch.sum((x - g) ** 2) + float(self.beta) * torch.sum(u ** 2)   # :39-41

    def _final_cost(self, x):
        return torch.sum((x - self._t(self.goal)) ** 2)               # :47


class Navigation(OracleEnv):
    """``tfmpc/envs/navigation/__init__.py:9-74`` (deterministic ``cec=True``)."""

    def __init__(self, goal, center, decay, low, high, dtype=np.float64):
        super().__init__(dtype)
        self.goal = np.asarray(goal, dtype=self.dtype).reshape(-1, 1)
        self.center = np.asarray(center, dtype=self.dtype).reshape(-1, self.goal.shape[0], 1)
        self.decay = np.asarray(decay, dtype=self.dtype).reshape(-1)
        self.state_size = self.action_size = self.goal.shape[0]
        self.action_space = Box(np.asarray(low).reshape(-1, 1), np.asarray(high).reshape(-1, 1),
                                self.goal.shape)                      # :21-24

    def _deceleration(self, x):
        delta = x.unsqueeze(0) - self._t(self.center)                 # [Z, n, 1]   :69
        distance = torch.sqrt(torch.sum(delta[..., 0] ** 2, dim=-1))  # :71
        lambdas = 2.0 / (1.0 + torch.exp(-self._t(self.decay) * distance)) - 1.0   # :73
        return torch.prod(lambdas)                                    # :74

    def _transition(self, x, u):
        return x + self._deceleration(x) * u                          # :36-43

    def _cost(self, x, u):
        return torch.sum((x - self._t(self.goal)) ** 2)               # :50-53

    def _final_cost(self, x):
        return torch.sum((x - self._t(self.goal)) ** 2)               # :56-59


class HVAC(OracleEnv):
    """``tfmpc/envs/hvac/__init__.py:8-149``"""

    CAP_AIR, COST_AIR, TEMP_AIR, TIME_DELTA = 1.006, 1.0, 40.0, 1.0   # :10-13
    PENALTY, SET_POINT_PENALTY = 20000.0, 10.0                        # :14-15

    def __init__(self, temp_outside, temp_hall, temp_lower_bound, temp_upper_bound,
                 R_outside, R_hall, R_wall, capacity, air_max, adj, adj_outside, adj_hall,
                 dtype=np.float64):
        super().__init__(dtype)
        col = lambda a: np.asarray(a, dtype=self.dtype).reshape(-1, 1)
        self.temp_outside, self.temp_hall = col(temp_outside), col(temp_hall)
        self.temp_lower_bound, self.temp_upper_bound = col(temp_lower_bound), col(temp_upper_bound)
        self.R_outside, self.R_hall = col(R_outside), col(R_hall)
        self.R_wall = np.asarray(R_wall, dtype=self.dtype)
        self.capacity, self.air_max = col(capacity), col(air_max)
        self.adj = np.asarray(adj, dtype=bool)
        self.adj_outside = np.asarray(adj_outside, dtype=bool).reshape(-1, 1)
        self.adj_hall = np.asarray(adj_hall, dtype=bool).reshape(-1, 1)
        self.state_size = self.action_size = self.temp_lower_bound.shape[0]   # :61-67
        self.action_space = Box(0.0, 1.0, (self.action_size, 1))      # :58-59

    def _transition(self, x, u):
        t = self._t
        air = u * t(self.air_max)                                     # :72
        heating = air * self.CAP_AIR * (self.TEMP_AIR - x)            # :74
        adj = t(np.logical_or(self.adj, self.adj.T))                  # :133-134
        between = torch.sum(-adj / t(self.R_wall) * (x - x.transpose(0, 1)),
                            dim=-1, keepdim=True)                     # :135-139
        outside = t(self.adj_outside) / t(self.R_outside) * (t(self.temp_outside) - x)   # :143-144
        hall = t(self.adj_hall) / t(self.R_hall) * (t(self.temp_hall) - x)               # :148-149
        return x + self.TIME_DELTA / t(self.capacity) * (heating + between + outside + hall)  # :80-88

    def _penalties(self, x):
        lo, hi = self._t(self.temp_lower_bound), self._t(self.temp_upper_bound)
        oob = self.PENALTY * (torch.relu(lo - x) + torch.relu(x - hi))         # :97-100
        sp = self.SET_POINT_PENALTY * torch.abs((lo + hi) / 2 - x)             # :101-105
        return oob + sp

    def _cost(self, x, u):
        air_cost = self.COST_AIR * (u * self._t(self.air_max))        # :94-96
        return torch.sum(air_cost + self._penalties(x))               # :107-110

    def _final_cost(self, x):
        return torch.sum(self._penalties(x))                          # :112-129


class Reservoir(OracleEnv):
    """``tfmpc/envs/reservoir/__init__.py:9-105`` (deterministic ``cec=True``)."""

    def __init__(self, max_res_cap, lower_bound, upper_bound, low_penalty, high_penalty,
                 set_point_penalty, downstream, rain_shape, rain_scale, dtype=np.float64):
        super().__init__(dtype)
        col = lambda a: np.asarray(a, dtype=self.dtype).reshape(-1, 1)
        self.max_res_cap = col(max_res_cap)
        self.lower_bound, self.upper_bound = col(lower_bound), col(upper_bound)
        self.low_penalty, self.high_penalty = col(low_penalty), col(high_penalty)
        self.set_point_penalty = col(set_point_penalty)
        self.downstream = np.asarray(downstream, dtype=self.dtype)
        self.rain_shape, self.rain_scale = col(rain_shape), col(rain_scale)
        self.state_size = self.action_size = self.lower_bound.shape[0]   # :39-45
        self.action_space = Box(0.0, 1.0, (self.action_size, 1))         # :36-37

    def _transition(self, x, u):
        t = self._t
        outflow = u * x                                                # :95
        vaporated = 0.5 * torch.sin(x / t(self.max_res_cap)) * x       # :87
        rainfall = t(self.rain_shape) * t(self.rain_scale)             # :100
        inflow = t(self.downstream).transpose(0, 1) @ outflow          # :91
        return x + rainfall + inflow - vaporated - outflow             # :56-60

    def _cost(self, x, u):
        t = self._t
        lo, hi = t(self.lower_bound), t(self.upper_bound)
        c1 = -t(self.low_penalty) * torch.relu(lo - x)                 # :70
        c2 = -t(self.high_penalty) * torch.relu(x - hi)                # :71
        c3 = -t(self.set_point_penalty) * torch.abs((lo + hi) / 2.0 - x)   # :72
        return torch.sum(c1 + c2 + c3)

    def _final_cost(self, x):
        return self._cost(x, None)                                     # :81-83


class LQEnv(OracleEnv):
    """Time-invariant LQ problem ``(F, f, C, c)`` of ``tfmpc/solvers/lqr.py:36-57``
    presented through the DiffEnv protocol, so that iLQR can be driven on the
    BASELINE.json headline shape (n=16, m=8) -- the reference's own envs all
    have ``action_size == state_size``.  Actions unbounded unless ``low`` / ``high`` are given."""

    def __init__(self, F, f, C, c, low=None, high=None, dtype=np.float64):
        super().__init__(dtype)
        self.F = np.asarray(F, dtype=self.dtype)
        self.f = np.asarray(f, dtype=self.dtype).reshape(-1, 1)
        self.C = np.asarray(C, dtype=self.dtype)
        self.c = np.asarray(c, dtype=self.dtype).reshape(-1, 1)
        self.state_size = self.F.shape[0]
        self.action_size = self.F.shape[1] - self.state_size
        self.action_space = Box(-np.inf if low is None else low, np.inf if high is None else high, (self.action_size, 1))

    def _transition(self, x, u):
        return self._t(self.F) @ torch.cat([x, u], dim=0) + self._t(self.f)

    def _cost(self, x, u):
        z = torch.cat([x, u], dim=0)
        return (0.5 * z.transpose(0, 1) @ self._t(self.C) @ z + z.transpose(0, 1) @ self._t(self.c)).reshape(())

    def _final_cost(self, x):
        n = self.state_size
        return (0.5 * x.transpose(0, 1) @ self._t(self.C[:n, :n]) @ x
                + x.transpose(0, 1) @ self._t(self.c[:n])).reshape(())
