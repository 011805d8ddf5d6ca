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
        # on_device: nothing is read back to the host inside __call__ (iteration counts stay device tensors, the start
        # actions of cold solves come from `start_actions`, prepared up front) -- what a captured episode needs
        # (runners.Runner.capture): no synchronisation and no host-to-device copy between the launches
        self.on_device = False
        self.start_actions = None   # on_device: per step the [B|-, steps, m, 1] start actions of a cold solve

    def reset(self):
        self._plan = None
        self.iterations = []

    def prepare_start_actions(self, batch):
        """The start actions of every (cold) re-solve of an episode, generated once: step t gets
        ``solver.random_actions(horizon - t, batch, seed + t)`` -- what ``__call__`` would draw."""
        self.start_actions = [self.solver.random_actions(self.horizon - t, batch, None if self.seed is None else self.seed + t)
                              for t in range(self.horizon if not self.warm_start else 1)]

    def __call__(self, state, timestep):
        steps_to_go = self.horizon - int(timestep)
        u_init = None
        if self.warm_start and self._plan is not None and self._plan.shape[-3] - 1 == steps_to_go:
            u_init = self._plan[..., 1:, :, :]
        seed = None if self.seed is None else self.seed + int(timestep)
        if u_init is None and self.on_device:
            u_init = self.start_actions[int(timestep)]
        out = self.solver.solve_device(state, steps_to_go, u_init=u_init, seed=seed)
        actions = out["actions"] if out["batched"] else out["actions"][0]
        self._plan = actions
        its = out["iterations"]
        if self.on_device:
            self.iterations.append(its if out["batched"] else its[0])
        else:
            self.iterations.append(its.cpu().numpy() if out["batched"] else int(its[0]))
        return actions[..., 0, :, :]            # first action, column vector(s) [.., m, 1]
