"""Differentiable-environment protocol -- drop-in for the reference's
``tfmpc/envs/diffenv.py`` (``TransitionApprox/CostApprox/FinalCostApprox`` :6-8,
``DiffEnv.get_linear_transition`` :13-32, ``get_quadratic_cost`` :34-83,
``get_quadratic_final_cost`` :85-101).

The reference differentiates ``transition`` / ``cost`` with TensorFlow autodiff.
Here every built-in env is described to the GPU library by a kind tag and its
parameter arrays (``struct TfmpcEnv`` in ``include/tfmpc_hip.h``) and the model
values AND their closed-form Jacobians / Hessians are evaluated by HIP kernels;
the methods below are thin launches of those kernels, so the same device code
serves the env API and the solver.
"""

import ctypes
from collections import namedtuple

import numpy as np
import torch

from tfmpc import _hip

TransitionApprox = namedtuple("TransitionApprox", "f f_x f_u")
CostApprox = namedtuple("CostApprox", "l l_x l_u l_xx l_uu l_ux l_xu")
FinalCostApprox = namedtuple("FinalCostApprox", "l l_x l_xx")


class Box:
    """The part of ``gym.spaces.Box`` the solver touches (``ilqr.py:47,51,136``):
    ``low``, ``high`` (float32, shape of the action column) and ``is_bounded()``."""

    def __init__(self, low, high, shape):
        shape = tuple(int(s) for s in shape)
        self.shape = shape
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float32), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float32), shape).copy()

    def is_bounded(self):
        return bool(np.all(np.isfinite(self.low)) and np.all(np.isfinite(self.high)))


def _f32(a, device):
    if isinstance(a, torch.Tensor):
        return a.detach().to(device=device, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float32)), device=device)


class DiffEnv:
    """Base of the built-in envs.  Subclasses set ``kind``, ``state_size``,
    ``action_size``, ``action_space`` and implement ``_params()`` returning the list
    of ``(array, per_instance_rank)`` parameter arrays in the order the C ABI
    documents (an array with one more axis than ``per_instance_rank`` is per-instance)."""

    kind = None
    n_zones = 0
    scalars = ()
    device = None

    # -- description handed to the C ABI ------------------------------------------
    def _device(self):
        if self.device is None:
            self.device = _hip.default_device()
        return self.device

    def _library(self):
        """The library whose entry points serve this env (a DeviceEnv answers with its companion library, envs/deviceenv.py)."""
        return _hip.require_gpu()

    def env_batch_size(self):
        sizes = {np.shape(a)[0] for a, rank in self._params() if np.ndim(a) == rank + 1}
        if len(sizes) > 1:
            raise ValueError("per-instance env parameters disagree on the batch size")
        return sizes.pop() if sizes else None

    def c_env(self):
        """(ctypes TfmpcEnv, keep-alive list of device tensors)."""
        dev = self._device()
        cached = getattr(self, "_c_env_cache", None)
        if cached is not None and cached[2] == dev:
            return cached[0], cached[1]
        env = _hip.TfmpcEnv()
        env.kind, env.n, env.m = int(self.kind), int(self.state_size), int(self.action_size)
        env.n_zones = int(self.n_zones)
        env.coupling_shift = int(getattr(self, "coupling_shift", 0))      # Reservoir: a chain topology, stated to the kernels (tfmpc_hip.h)
        env.bounded = int(self.action_space.is_bounded())
        env.any_finite_bound = int(bool(np.any(np.isfinite(self.action_space.low)) or
                                        np.any(np.isfinite(self.action_space.high))))
        keep = []
        low = _f32(self.action_space.low.reshape(-1), dev)
        high = _f32(self.action_space.high.reshape(-1), dev)
        keep += [low, high]
        env.low, env.high = low.data_ptr(), high.data_ptr()
        for i, (arr, rank) in enumerate(self._params()):
            t = _f32(arr, dev)
            keep.append(t)
            env.p[i] = t.data_ptr()
            env.stride[i] = t[0].numel() if t.dim() == rank + 1 else 0
        for i, v in enumerate(self.scalars):
            env.scalar[i] = float(v)
        self._c_env_cache = (env, keep, dev)
        return env, keep

    # -- shape helpers ---------------------------------------------------------------
    def _cols(self, a, size):
        """-> (tensor [B, size], batched?)"""
        t = _f32(a, self._device())
        if t.dim() >= 2 and t.shape[-1] == 1 and t.shape[-2] == size:
            t = t.squeeze(-1)
        if t.shape[-1] != size or t.dim() not in (1, 2):
            raise ValueError(f"expected [{size},1] or [B,{size},1], got {tuple(np.shape(a))}")
        return (t.unsqueeze(0), False) if t.dim() == 1 else (t, True)

    def _step(self, state, action):
        lib = self._library()
        n, m = self.state_size, self.action_size
        x, bx = self._cols(state, n)
        if action is None:
            u, bu = torch.zeros((x.shape[0], m), device=x.device), False
        else:
            u, bu = self._cols(action, m)
        B = max(x.shape[0], u.shape[0])
        x = x.expand(B, n).contiguous()
        u = u.expand(B, m).contiguous()
        env, keep = self.c_env()
        if self.env_batch_size() not in (None, B):
            raise ValueError("batch does not match the env's per-instance parameters")
        states = torch.empty((B, 2, n), device=x.device)
        costs = torch.empty((B, 2), device=x.device)
        rc = lib.tfmpc_ilqr_rollout_f32(ctypes.byref(env), B, 1, _hip.ptr(x), _hip.ptr(u), _hip.ptr(states),
                                        _hip.ptr(costs), _hip.stream())
        _hip.check(rc, "tfmpc_ilqr_rollout_f32")
        return states[:, 1], costs[:, 0], (bx or bu)

    # -- reference protocol ----------------------------------------------------------
    def transition(self, state, action, batch=False, cec=True):
        nxt, _, batched = self._step(state, action)
        nxt = nxt.unsqueeze(-1)
        if not cec:        # stochastic dynamics: deterministic kernel + the env's noise model (gymenv.py)
            if not hasattr(self, "_noise"):
                raise NotImplementedError(f"{type(self).__name__} has no stochastic dynamics")
            if getattr(self, "_generator", None) is None:
                self.seed(None)
            nxt = nxt + self._noise(nxt)
        return nxt if batched else nxt[0]

    def cost(self, state, action, batch=False):
        _, c, batched = self._step(state, action)
        return c if batched else c[0]

    def final_cost(self, state, batch=False):
        lib = self._library()
        x, batched = self._cols(state, self.state_size)
        x = x.contiguous()
        env, keep = self.c_env()
        B = x.shape[0]
        states = torch.empty((B, 1, self.state_size), device=x.device)
        costs = torch.empty((B, 1), device=x.device)
        rc = lib.tfmpc_ilqr_rollout_f32(ctypes.byref(env), B, 0, _hip.ptr(x), None, _hip.ptr(states), _hip.ptr(costs),
                                        _hip.stream())
        _hip.check(rc, "tfmpc_ilqr_rollout_f32")
        return costs[:, 0] if batched else costs[0, 0]

    def _derivatives(self, state, action, want):
        """One derivatives launch with T = 1 per row of the (time-)batch."""
        lib = self._library()
        n, m = self.state_size, self.action_size
        x, bx = self._cols(state, n)
        u, bu = self._cols(action, m) if action is not None else (torch.zeros((x.shape[0], m), device=x.device), False)
        B = max(x.shape[0], u.shape[0])
        u = u.expand(B, m).contiguous()
        states = torch.stack([x.expand(B, n), x.expand(B, n)], dim=1).contiguous()      # [B, T+1=2, n]
        env, keep = self.c_env()
        dev = x.device
        shapes = dict(f=(B, n, 1), f_x=(B, n, n), f_u=(B, n, m), l=(B,), l_x=(B, n, 1), l_u=(B, m, 1), l_xx=(B, n, n),
                      l_uu=(B, m, m), l_ux=(B, m, n), l_xu=(B, n, m), fl=(B,), fl_x=(B, n, 1), fl_xx=(B, n, n))
        out = {k: (torch.empty(shapes[k], device=dev) if k in want else None) for k in shapes}
        order = ("f", "f_x", "f_u", "l", "l_x", "l_u", "l_xx", "l_uu", "l_ux", "l_xu", "fl", "fl_x", "fl_xx")
        rc = lib.tfmpc_ilqr_derivatives_f32(ctypes.byref(env), B, 1, _hip.ptr(states), _hip.ptr(u),
                                            *[_hip.ptr(out[k]) for k in order], _hip.stream())
        _hip.check(rc, "tfmpc_ilqr_derivatives_f32")
        batched = bx or bu
        return {k: (v if batched else v[0]) for k, v in out.items() if v is not None}

    def get_linear_transition(self, state, action, batch=True):
        o = self._derivatives(state, action, ("f", "f_x", "f_u"))
        return TransitionApprox(o["f"], o["f_x"], o["f_u"])

    def get_quadratic_cost(self, state, action, batch=True):
        o = self._derivatives(state, action, ("l", "l_x", "l_u", "l_xx", "l_uu", "l_ux", "l_xu"))
        return CostApprox(o["l"], o["l_x"], o["l_u"], o["l_xx"], o["l_uu"], o["l_ux"], o["l_xu"])

    def get_quadratic_final_cost(self, state):
        o = self._derivatives(state, None, ("fl", "fl_x", "fl_xx"))
        return FinalCostApprox(o["fl"], o["fl_x"], o["fl_xx"])
