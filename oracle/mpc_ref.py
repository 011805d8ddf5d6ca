"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's online (receding-horizon) MPC loop with the random
draws INJECTED, so that batched GPU episodes can be compared with it step by step:

* ``MPCRef``     -- ``/root/reference/tfmpc/agents/mpc.py:4-15``
* ``GymEnvRef``  -- ``/root/reference/tfmpc/envs/gymenv.py:4-41`` over an ``envs_ref`` env, stepping with the env's
                    stochastic dynamics (``cec=False``): Navigation adds a truncated-normal displacement
                    (``envs/navigation/__init__.py:42-45``), Reservoir replaces the mean rainfall ``shape * scale``
                    by a Gamma draw (``envs/reservoir/__init__.py:97-105``); the draws come from ``samples[t]``
* ``RunnerRef``  -- ``/root/reference/tfmpc/runners/__init__.py:8-49``

The reference cold-starts every re-solve from fresh random actions (``ilqr.py:218`` via ``start``); here they are the
injected ``u_inits[timestep]``.  TensorFlow's RNG streams cannot be reproduced outside TensorFlow, which is why both
sources of randomness are arguments.  PARITY UNPINNED (the reference's tests hold no episode).
"""

import numpy as np

from . import envs_ref


def stochastic_transition(env, state, action, sample):
    """``transition(state, action, cec=False)`` with the env's random draw given."""
    dt = env.dtype
    state = np.asarray(state, dtype=dt)
    sample = np.asarray(sample, dtype=dt).reshape(state.shape)
    if isinstance(env, envs_ref.Navigation):
        return np.asarray(env.transition(state, action), dtype=dt) + sample           # navigation/__init__.py:42-45
    if isinstance(env, envs_ref.Reservoir):
        mean = env.rain_shape * env.rain_scale                                         # reservoir/__init__.py:100
        # next_state = rlevel + rainfall + inflow - vaporated - outflow (:56-60): only the rainfall term changes
        return np.asarray(env.transition(state, action), dtype=dt) - mean + sample
    raise TypeError(f"{type(env).__name__} has no stochastic mode in the reference (HVAC is not a GymEnv)")


class GymEnvRef:
    """gymenv.py:4-41"""

    def __init__(self, env, samples):
        self.env, self.samples = env, samples
        self._t, self._state = None, None

    def setup(self, initial_state, horizon):                   # :11-13
        self.initial_state, self.horizon = np.asarray(initial_state, dtype=self.env.dtype).reshape(-1, 1), horizon

    def reset(self):                                           # :27-32
        self._t, self._state = 0, self.initial_state
        return self._state

    def step(self, action):                                    # :15-25
        self._t += 1
        next_state = stochastic_transition(self.env, self._state, action, self.samples[self._t - 1])
        cost = self.env.cost(self._state, action)
        done = self._t == self.horizon
        self._state = next_state
        return next_state, cost, done, {}

    def final_cost(self, state):
        return self.env.final_cost(state)


class MPCRef:
    """agents/mpc.py:4-15 around an ``ilqr_ref.ILQRRef``."""

    def __init__(self, solver, horizon, u_inits):
        self.solver, self.horizon, self.u_inits = solver, horizon, u_inits
        self.iterations = []

    def __call__(self, state, timestep):
        steps_to_go = self.horizon - timestep                  # :11
        states, actions, costs, iteration = self.solver.solve(state, steps_to_go, u_init=self.u_inits[timestep])   # :12
        self.iterations.append(iteration)
        return np.asarray(actions[0], dtype=self.solver.dtype).reshape(-1, 1)          # :13-15 first action, column


class RunnerRef:
    """runners/__init__.py:8-49"""

    def __init__(self, env, agent):
        self.env, self.agent = env, agent

    def run(self):                                             # :14-43
        state = self.env.reset()
        timestep, done = 0, False
        states, actions, costs = [state], [], []
        while not done:
            action = self.agent(state, timestep)
            next_state, cost, done, _ = self.env.step(action)
            state = next_state
            timestep = self.env._t
            states.append(state)
            actions.append(action)
            costs.append(cost)
        costs.append(self.env.final_cost(state))
        return np.stack(states)[..., 0], np.stack(actions)[..., 0], np.asarray(costs, dtype=np.float64)
