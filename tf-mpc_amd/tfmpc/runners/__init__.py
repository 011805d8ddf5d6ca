"""Episode loop env <-> agent -- drop-in for the reference's ``tfmpc/runners/__init__.py:8-49``,
batched over ``B`` independent episodes when the env's initial state is ``[B,n,1]``."""

import contextlib

import torch

from tfmpc.utils import trajectory


class Runner:

    def __init__(self, env, agent):
        self.env = env
        self.agent = agent

    def run(self, mode=None):
        state = self.env.reset()
        if hasattr(self.agent, "reset"):
            self.agent.reset()
        timestep = 0
        done = False
        states, actions, costs = [state], [], []
        while not done:
            action = self.agent(state, timestep)
            next_state, cost, done, info = self.env.step(action)
            if mode is not None:
                self.env.render(mode)
            state = next_state
            timestep = self.env._t
            states.append(state)
            actions.append(action)
            costs.append(cost)
        costs.append(self.env.final_cost(state, batch=state.dim() == 3))
        tdim = state.dim() - 2          # time axis goes after the batch axis
        return trajectory.Trajectory(torch.stack(states, dim=tdim), torch.stack(actions, dim=tdim),
                                     torch.stack(costs, dim=tdim))

    @contextlib.contextmanager
    def __call__(self, initial_state, horizon):
        self.env.setup(initial_state, horizon)
        yield self
        self.env.close()
