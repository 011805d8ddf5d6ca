"""Arbitrary differentiable environments.

The reference accepts ANY ``transition`` / ``cost`` / ``final_cost`` written in TensorFlow ops
and differentiates them with a gradient tape (``tfmpc/envs/diffenv.py:13-101``).  The built-in
envs of this package carry closed-form derivatives inside HIP kernels instead; ``TorchEnv``
restores the reference's generality: the model is given as three torch functions of ONE
instance (``x[n], u[m] -> x'[n]``, ``-> scalar``, ``x[n] -> scalar``) and the linear/quadratic
models come from ``torch.func`` (``vmap`` over batch and time of ``jacrev`` / ``jacfwd``), on the
GPU.  ``iLQR`` then runs the reference's loop with the Riccati backward pass in the HIP kernel
(``tfmpc_ilqr_backward_f32``, materialised models) and the rollouts as batched torch ops.

Round 6: that host-driven loop is the FALLBACK.  ``iLQR(TorchEnv(...))`` first tries ``to_device_env()`` -- the functions
traced and translated into device source (``envs/fxsource.py``), compiled into the fused kernels, two orders of magnitude
faster -- and only an env the translator cannot take (an unsupported operation, a Python branch on the state, no hipcc)
stays here (``iLQR(...).compile_error`` says why; ``TorchEnv(..., auto_compile=False)`` / ``iLQR(env, compile_env=False)`` ask for it).
"""

import numpy as np
import torch
from torch.func import jacfwd, jacrev, vmap

from tfmpc import _hip
from tfmpc.envs.diffenv import Box, CostApprox, FinalCostApprox, TransitionApprox


class TorchEnv:
    kind = None          # not one of the device-resident env kinds
    auto_compile = True  # iLQR(env) translates the functions to device source when it can (to_device_env); False keeps the host-driven loop

    def __init__(self, transition_fn, cost_fn, final_cost_fn, state_size, action_size, low=None, high=None,
                 device=None, auto_compile=None):
        if auto_compile is not None:
            self.auto_compile = bool(auto_compile)
        self._f, self._l, self._lf = transition_fn, cost_fn, final_cost_fn
        self.state_size, self.action_size = int(state_size), int(action_size)
        low = -np.inf if low is None else low
        high = np.inf if high is None else high
        self.obs_space = Box(-np.inf, np.inf, (self.state_size, 1))
        self.action_space = Box(low, high, (self.action_size, 1))
        self.device = torch.device(device) if device is not None else None

    def _device(self):
        if self.device is None:
            self.device = _hip.default_device()
        return self.device

    def to_device_env(self, trace_device=None):
        """The SAME env at hot-path speed: the three torch functions are traced (``torch.fx`` ``make_fx``) and translated into the device
        templates of a ``tfmpc.envs.deviceenv.DeviceEnv`` (``envs/fxsource.py``: what is supported, the subgradient conventions, and the
        error an unsupported operation or data-dependent Python control flow raises), compiled with hipcc at first use into the fused
        wave-per-instance kernels -- derivatives by dual numbers on the device instead of ``torch.func`` on the host.  Tensor constants the
        functions close over become the env's ``params``.  ``trace_device``: where the example inputs of the trace live (default: CPU, and
        the GPU if the functions' constants turn out to live there)."""
        from tfmpc.envs import fxsource
        from tfmpc.envs.deviceenv import DeviceEnv
        devices = [trace_device] if trace_device is not None else ["cpu", self._device()]
        error = None
        for dev in devices:
            try:
                source, params, info = fxsource.translate_ex(self._f, self._l, self._lf, self.state_size, self.action_size, device=dev)
                break
            except RuntimeError as exc:                       # constants on another device than the example inputs
                if "device" not in str(exc):
                    raise
                error = exc
        else:
            raise error
        # a cost the translator found piecewise affine (all second derivatives exactly zero): with bounded actions the reference's backward pass
        # only ever takes its bang-bang branch (ilqr.py:137-141) -- the companion library is then built on the costate form of the kernels
        env = DeviceEnv(source, self.state_size, self.action_size, params=params, low=self.action_space.low, high=self.action_space.high,
                        zero_cost_hessian=info["cost_is_piecewise_linear"])
        env.device = self.device
        return env

    def env_batch_size(self):
        return None

    # ---- batched evaluation: leading axes are flattened, functions vmapped -------------------
    def _flat(self, a, size):
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float32))
        t = t.to(device=self._device(), dtype=torch.float32)
        if t.dim() >= 2 and t.shape[-1] == 1 and t.shape[-2] == size:
            t = t.squeeze(-1)
        if t.shape[-1] != size:
            raise ValueError(f"expected a trailing axis of {size}, got {tuple(t.shape)}")
        return t.reshape(-1, size), t.shape[:-1]

    def step_flat(self, x, u):
        """x[R,n], u[R,m] -> x'[R,n], cost[R]"""
        return vmap(self._f)(x, u), vmap(self._l)(x, u)

    def final_cost_flat(self, x):
        return vmap(self._lf)(x)

    def transition(self, state, action, batch=False, cec=True):
        x, lead = self._flat(state, self.state_size)
        u, _ = self._flat(action, self.action_size)
        return vmap(self._f)(x, u).reshape(*lead, self.state_size, 1)

    def cost(self, state, action, batch=False):
        x, lead = self._flat(state, self.state_size)
        u, _ = self._flat(action, self.action_size)
        return vmap(self._l)(x, u).reshape(lead)

    def final_cost(self, state, batch=False):
        x, lead = self._flat(state, self.state_size)
        return vmap(self._lf)(x).reshape(lead)

    # ---- diffenv.py:13-101 ------------------------------------------------------------------------
    def get_linear_transition(self, state, action, batch=True):
        x, lead = self._flat(state, self.state_size)
        u, _ = self._flat(action, self.action_size)
        n, m = self.state_size, self.action_size
        f = vmap(self._f)(x, u)
        f_x, f_u = vmap(jacrev(self._f, argnums=(0, 1)))(x, u)
        return TransitionApprox(f.reshape(*lead, n, 1), f_x.reshape(*lead, n, n), f_u.reshape(*lead, n, m))

    def get_quadratic_cost(self, state, action, batch=True):
        x, lead = self._flat(state, self.state_size)
        u, _ = self._flat(action, self.action_size)
        n, m = self.state_size, self.action_size
        l = vmap(self._l)(x, u)
        l_x, l_u = vmap(jacrev(self._l, argnums=(0, 1)))(x, u)
        (l_xx, l_xu), (l_ux, l_uu) = vmap(jacfwd(jacrev(self._l, argnums=(0, 1)), argnums=(0, 1)))(x, u)
        return CostApprox(l.reshape(lead), l_x.reshape(*lead, n, 1), l_u.reshape(*lead, m, 1), l_xx.reshape(*lead, n, n),
                          l_uu.reshape(*lead, m, m), l_ux.reshape(*lead, m, n), l_xu.reshape(*lead, n, m))

    def get_quadratic_final_cost(self, state):
        x, lead = self._flat(state, self.state_size)
        n = self.state_size
        l = vmap(self._lf)(x)
        l_x = vmap(jacrev(self._lf))(x)
        l_xx = vmap(jacfwd(jacrev(self._lf)))(x)
        return FinalCostApprox(l.reshape(lead), l_x.reshape(*lead, n, 1), l_xx.reshape(*lead, n, n))
