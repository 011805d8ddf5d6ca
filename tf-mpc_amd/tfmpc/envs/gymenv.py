"""Episode stepping for online MPC -- drop-in for the reference's ``tfmpc/envs/gymenv.py:4-41``
(``setup / reset / step / render / close / seed``), batched: ``initial_state`` may be
``[B,n,1]`` and then every call steps ``B`` independent episodes.

``step`` applies the env's STOCHASTIC dynamics (``cec=False`` in the reference,
``gymenv.py:18``): the deterministic transition runs in the HIP env kernel and the env's
noise model is added on the device with torch's generator -- TensorFlow's RNG stream cannot
be reproduced, so episodes are reproducible per ``seed`` here, not against the reference."""

import torch

from tfmpc import _hip


class GymEnv:
    stochastic = True          # False: step() uses the certainty-equivalent dynamics (tests, ablations)

    def _gym_init(self):
        self._t = None
        self._state = None
        self._info = {}
        self._generator = None
        self._injected = None

    def inject_noise(self, samples):
        """Replace the env's random draws by a fixed sequence: ``samples[t]`` is the raw draw of the env's noise
        model for the step that ends at time t + 1 (Navigation: the truncated-normal displacement,
        ``navigation/__init__.py:45``; Reservoir: the Gamma rainfall, ``reservoir/__init__.py:101-104``), shaped
        like the state.  ``None`` returns to drawing.  This is how episodes are compared with the oracle's
        restatement of the loop: TensorFlow's RNG stream cannot be reproduced, an injected sequence can."""
        self._injected = None if samples is None else [
            torch.as_tensor(s, dtype=torch.float32).to(self._device()) for s in samples]

    def setup(self, initial_state, horizon):
        self.initial_state = initial_state
        self.horizon = int(horizon)

    def seed(self, seed=None):
        dev = self._device()
        self._generator = torch.Generator(device=dev)
        if seed is not None:
            self._generator.manual_seed(int(seed))

    def reset(self):
        self._t = 0
        x = self.initial_state
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(x, dtype=torch.float32)
        self._state = x.to(device=self._device(), dtype=torch.float32)
        self._info = {}
        return self._state

    def _noise(self, state):
        """Additive difference between a stochastic and the certainty-equivalent step (drawn)."""
        return torch.zeros_like(state)

    def _noise_from_sample(self, sample, state):
        """The same difference for an injected raw draw of the env's noise model."""
        return sample.expand_as(state)

    def step(self, action):
        self._t += 1
        batched = self._state.dim() == 3
        next_state = self.transition(self._state, action, batch=batched)
        if self.stochastic:
            if getattr(self, "_generator", None) is None:
                self.seed(None)
            if self._injected is not None:
                next_state = next_state + self._noise_from_sample(self._injected[self._t - 1], next_state)
            else:
                next_state = next_state + self._noise(next_state)
        cost = self.cost(self._state, action, batch=batched)
        done = self._t == self.horizon
        self._state = next_state
        return next_state, cost, done, self._info

    def render(self, mode="human"):
        pass

    def close(self):
        pass
