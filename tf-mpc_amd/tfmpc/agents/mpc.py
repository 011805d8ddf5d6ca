"""Receding-horizon controller -- drop-in for the reference's ``tfmpc/agents/mpc.py:4-15``:
at every step re-solve iLQR over the remaining horizon from the current state and apply the
first action.  Batched: ``state[B,n,1]`` -> ``action[B,m,1]``, one fused-solve launch per step.

The reference cold-starts every solve from a fresh random trajectory (``ilqr.py:218``).
``warm_start=True`` (build addition) starts step t+1 from the tail of the
plan of step t instead, which is what makes online MPC cheap: the shifted plan is already
near-optimal, so most re-solves converge in one or two iterations."""

import torch


class MPC:

    def __init__(self, solver, horizon, warm_start=False, seed=None):
        self.solver = solver
        self.horizon = int(horizon)
        self.warm_start = warm_start
        self.seed = seed
        self._plan = None           # actions of the previous solve, [B|-, steps, m, 1]
        self.iterations = []        # per step: iterations the solve took (int or array[B])

    def reset(self):
        self._plan = None
        self.iterations = []

    def __call__(self, state, timestep):
        steps_to_go = self.horizon - int(timestep)
        u_init = None
        if self.warm_start and self._plan is not None and self._plan.shape[-3] - 1 == steps_to_go:
            u_init = self._plan[..., 1:, :, :]
        seed = None if self.seed is None else self.seed + int(timestep)
        out = self.solver.solve_device(state, steps_to_go, u_init=u_init, seed=seed)
        actions = out["actions"] if out["batched"] else out["actions"][0]
        self._plan = actions
        its = out["iterations"]
        self.iterations.append(its.cpu().numpy() if out["batched"] else int(its[0]))
        return actions[..., 0, :, :]            # first action, column vector(s) [.., m, 1]
