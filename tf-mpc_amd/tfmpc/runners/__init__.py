"""Closed-loop episode driver: the env is stepped under an agent for the horizon configured by
``env.setup`` and the visited (state, action, cost) triples come back as a ``Trajectory``.
Same surface as the reference's ``tfmpc/runners/__init__.py:8-49`` (``Runner(env, agent).run()``
and the ``with runner(x0, T) as r`` form); here ``B`` episodes advance together when the
initial state is ``[B,n,1]`` -- one fused iLQR launch per control step for all of them."""

import contextlib

import torch

from tfmpc.utils.trajectory import Trajectory


class Runner:

    def __init__(self, env, agent):
        self.env, self.agent = env, agent

    @contextlib.contextmanager
    def __call__(self, initial_state, horizon):
        self.env.setup(initial_state, horizon)
        try:
            yield self
        finally:
            self.env.close()

    def run(self, mode=None):
        env, agent = self.env, self.agent
        x = env.reset()
        reset_agent = getattr(agent, "reset", None)
        if callable(reset_agent):
            reset_agent()
        batched = x.dim() == 3
        time_axis = 1 if batched else 0
        visited, applied, paid = [x], [], []
        if env.horizon < 1:
            raise ValueError("Runner.run needs a horizon of at least one step")
        finished = False
        while not finished:
            u = agent(x, env._t)                       # plan over the remaining horizon, apply the first action
            x, stage_cost, finished, _ = env.step(u)
            if mode is not None:
                env.render(mode)
            visited.append(x)
            applied.append(u)
            paid.append(stage_cost)
        paid.append(env.final_cost(x, batch=batched))
        return Trajectory(torch.stack(visited, dim=time_axis), torch.stack(applied, dim=time_axis),
                          torch.stack(paid, dim=time_axis))
