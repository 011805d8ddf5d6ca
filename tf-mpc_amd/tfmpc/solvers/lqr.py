"""Finite-horizon LQR on MI355X -- drop-in for the reference's
``tfmpc/solvers/lqr.py`` (``LQR``: ``__init__`` :18-22, ``transition`` :36-39,
``cost`` :41-47, ``final_cost`` :49-57, ``backward`` :59-129, ``forward``
:131-161, ``solve`` :163-166, ``dump``/``load`` :168-181).

Differences from the reference, all additive:

* tensors are torch (fp32, on the ROCm device) / numpy instead of ``tf.Tensor``;
* ``F, f, C, c`` and ``x0`` may each carry one leading batch axis ``B``: the call
  then solves ``B`` independent problems in ONE kernel launch (an operand without
  the axis is shared by all instances);
* ``backward`` / ``forward`` / ``solve`` execute as hand-written gfx950 kernels
  through the C ABI (``include/tfmpc_hip.h``).  No CPU fallback.
* ``C`` is expected to be symmetric (every problem the reference builds is:
  ``make_lqr`` draws ``make_spd_matrix``, ``tfmpc/envs/__init__.py:9-18``) with ``C_uu``
  positive definite; the fast kernels use that.  A ``C`` that is NOT symmetric is detected
  once at construction and solved by the ``*_general_f32`` entry points, which restate the
  reference's recursion term by term (``lqr.py:74-105``: ``Q_ux`` and ``Q_xu`` separately,
  general inverse, no symmetrisation) -- slower, same results as the reference.
"""

import json

import numpy as np
import torch

from tfmpc import _hip
from tfmpc.utils import trajectory


class Policy(list):
    """``policy[t] == (K_t, k_t)`` like the reference's list, plus the stacked
    device tensors ``K[(B,)T,m,n]``, ``k[(B,)T,m,1]`` the kernels read."""

    def __init__(self, K, k):
        tdim = K.dim() - 3
        super().__init__((K.select(tdim, t), k.select(tdim, t)) for t in range(K.shape[tdim]))
        self.K, self.k = K, k


class ValueFn(list):
    """``value_fn[t] == (V_t, v_t, const_t)`` plus the stacked tensors."""

    def __init__(self, V, v, const):
        tdim = V.dim() - 3
        super().__init__((V.select(tdim, t), v.select(tdim, t), const.select(tdim, t))
                         for t in range(V.shape[tdim]))
        self.V, self.v, self.const = V, v, const


def _as_f32(a, device):
    if isinstance(a, torch.Tensor):
        return a.detach().to(device=device, dtype=torch.float32)
    return torch.as_tensor(np.asarray(a, dtype=np.float32), device=device)


def _as_column(t, size):
    """``[..., size]`` -> ``[..., size, 1]``; column vectors pass through."""
    if t.dim() >= 2 and t.shape[-1] == 1 and t.shape[-2] == size:
        return t
    return t.unsqueeze(-1)


class LQR:

    def __init__(self, F, f, C, c, device=None, symmetric=None):
        """``symmetric``: ``None`` (default) checks once whether ``C`` is symmetric -- the fast kernels' precondition,
        include/tfmpc_hip.h -- which costs two temporaries and a host read-back; ``True`` / ``False`` states it and
        skips the check (construction is then free of synchronisation, e.g. inside a stream capture)."""
        self.device = torch.device(device) if device is not None else _hip.default_device()
        F, f, C, c = (_as_f32(a, self.device) for a in (F, f, C, c))
        n, d = F.shape[-2], F.shape[-1]
        f, c = _as_column(f, n), _as_column(c, d)
        if not (0 < n < d):
            raise ValueError(f"F must be [n, n+m] with m > 0, got {tuple(F.shape)}")
        for name, t, shape in (("f", f, (n, 1)), ("C", C, (d, d)), ("c", c, (d, 1))):
            if tuple(t.shape[-2:]) != shape:
                raise ValueError(f"{name} must end in shape {shape}, got {tuple(t.shape)}")
        batches = {t.shape[0] for t, nd in ((F, 3), (f, 3), (C, 3), (c, 3)) if t.dim() == nd}
        if any(t.dim() not in (2, 3) for t in (F, f, C, c)) or len(batches) > 1:
            raise ValueError("F, f, C, c take at most one leading batch axis of a common size")
        self.F, self.f, self.C, self.c = (t.contiguous() for t in (F, f, C, c))
        self.batch_size = batches.pop() if batches else None
        self.last_status = None
        # asymmetry beyond fp32 rounding of a symmetric matrix -> the reference's term-by-term recursion
        if symmetric is not None:
            self.symmetric_cost = bool(symmetric)
        elif self.C.numel() == 0:                      # an empty shard (parallel.shard of a small batch)
            self.symmetric_cost = True
        else:
            asym = (self.C - self.C.transpose(-1, -2)).abs().amax()
            self.symmetric_cost = bool(asym <= 1e-6 * self.C.abs().amax())
        self._suffix = "_f32" if self.symmetric_cost else "_general_f32"

    # -- reference properties (lqr.py:24-34) -----------------------------------
    @property
    def n_dim(self):
        return self.F.shape[-1]

    @property
    def state_size(self):
        return self.F.shape[-2]

    @property
    def action_size(self):
        return self.n_dim - self.state_size

    # -- single-step model (lqr.py:36-57); plain tensor ops, not the hot path ---
    def transition(self, x, u):
        z = torch.cat([_as_f32(x, self.device), _as_f32(u, self.device)], dim=-2)
        return self.F @ z + self.f

    def cost(self, x, u):
        z = torch.cat([_as_f32(x, self.device), _as_f32(u, self.device)], dim=-2)
        zt = z.transpose(-1, -2)
        return 0.5 * (zt @ self.C) @ z + zt @ self.c

    def final_cost(self, x):
        x = _as_f32(x, self.device)
        n = self.state_size
        xt = x.transpose(-1, -2)
        return 0.5 * (xt @ self.C[..., :n, :n]) @ x + xt @ self.c[..., :n, :]

    # -- helpers -----------------------------------------------------------------
    def _operands(self):
        """(tensor, batch stride in elements) for F, f, C, c."""
        out = []
        for t in (self.F, self.f, self.C, self.c):
            out.append((t, t[0].numel() if t.dim() == 3 else 0))
        return out

    def _ptr_args(self):
        args = []
        for t, stride in self._operands():
            args += [_hip.ptr(t), stride]
        return args

    def _resolve_batch(self, x0=None):
        B = self.batch_size
        if x0 is not None and x0.dim() == 3:
            if B is not None and x0.shape[0] != B:
                raise ValueError(f"x0 batch {x0.shape[0]} != problem batch {B}")
            B = x0.shape[0]
        return B

    def _prep_x0(self, x0):
        x0 = _as_f32(x0, self.device)
        n = self.state_size
        x0 = _as_column(x0, n)
        if tuple(x0.shape[-2:]) != (n, 1) or x0.dim() not in (2, 3):
            raise ValueError(f"x0 must be [n,1] or [B,n,1] with n={n}, got {tuple(x0.shape)}")
        return x0.contiguous()

    # -- lqr.py:59-129 -------------------------------------------------------------
    def backward(self, T):
        lib = _hip.require_gpu()
        T = int(T)
        n, m = self.state_size, self.action_size
        B = self.batch_size
        Bk = B or 1
        dev = self.device
        K = torch.empty((Bk, T, m, n), device=dev)
        k = torch.empty((Bk, T, m, 1), device=dev)
        V = torch.empty((Bk, T, n, n), device=dev)
        v = torch.empty((Bk, T, n, 1), device=dev)
        const = torch.empty((Bk, T, 1, 1), device=dev)
        status = torch.zeros((Bk,), dtype=torch.int32, device=dev)
        rc = getattr(lib, "tfmpc_lqr_backward" + self._suffix)(Bk, n, m, T, *self._ptr_args(),
                                        _hip.ptr(K), _hip.ptr(k), _hip.ptr(V), _hip.ptr(v), _hip.ptr(const),
                                        _hip.ptr(status), _hip.stream())
        _hip.check(rc, "tfmpc_lqr_backward_f32")
        self.last_status = status
        if B is None:
            K, k, V, v, const = K[0], k[0], V[0], v[0], const[0]
        return Policy(K, k), ValueFn(V, v, const)

    # -- lqr.py:131-161 ------------------------------------------------------------
    def forward(self, policy, x0, T):
        lib = _hip.require_gpu()
        T = int(T)
        n, m = self.state_size, self.action_size
        x0 = self._prep_x0(x0)
        if isinstance(policy, Policy):
            K, k = policy.K, policy.k
        else:       # a plain list of (K_t, k_t) as the reference returns
            tdim = _as_f32(policy[0][0], self.device).dim() - 2
            K = torch.stack([_as_f32(p[0], self.device) for p in policy], dim=tdim)
            k = torch.stack([_as_f32(p[1], self.device) for p in policy], dim=tdim)
        K, k = K.contiguous(), k.contiguous()
        if K.shape[-3] < T:
            raise ValueError(f"policy has {K.shape[-3]} steps, horizon is {T}")
        pol_batched = K.dim() == 4
        B = self._resolve_batch(x0)
        if pol_batched:
            if B is not None and K.shape[0] != B:
                raise ValueError("policy batch does not match")
            B = K.shape[0]
        Bk = B or 1
        if x0.dim() == 2:
            x0 = x0.unsqueeze(0).expand(Bk, n, 1).contiguous()
        dev = self.device
        states = torch.empty((Bk, T + 1, n, 1), device=dev)
        actions = torch.empty((Bk, T, m, 1), device=dev)
        costs = torch.empty((Bk, T + 1, 1, 1), device=dev)
        sK = K[0].numel() if pol_batched else 0
        sk = k[0].numel() if pol_batched else 0
        rc = getattr(lib, "tfmpc_lqr_forward" + self._suffix)(Bk, n, m, T, *self._ptr_args(),
                                       _hip.ptr(K), sK, _hip.ptr(k), sk, _hip.ptr(x0),
                                       _hip.ptr(states), _hip.ptr(actions), _hip.ptr(costs), _hip.stream())
        _hip.check(rc, "tfmpc_lqr_forward_f32")
        if B is None:
            states, actions, costs = states[0], actions[0], costs[0]
        return states, actions, costs

    # -- fused backward + forward, device tensors in/out ----------------------------
    def solve_device(self, x0, T, want_policy=False, want_value=False, workspace=None, storage_bf16=False):
        """One kernel launch for ``B`` solves; returns a dict of device tensors
        (``states[B,T+1,n,1]``, ``actions[B,T,m,1]``, ``costs[B,T+1,1,1]``,
        ``status[B]`` and, on request, ``K, k, V, v, const``).  Never synchronises.

        ``storage_bf16=True``: the requested policy / value-function tensors come back as ``torch.bfloat16`` (half the
        bytes of the path's largest outputs, lqr.py:107-129; each value = the fp32 result rounded to nearest even,
        ``tfmpc_lqr_solve_bf16out_f32``); the trajectory stays fp32 and is the one the fp32 gains give."""
        lib = _hip.require_gpu()
        T = int(T)
        n, m = self.state_size, self.action_size
        x0 = self._prep_x0(x0)
        B = self._resolve_batch(x0)
        Bk = B or 1
        if x0.dim() == 2:
            x0 = x0.unsqueeze(0).expand(Bk, n, 1).contiguous()
        dev = self.device
        out = dict(states=torch.empty((Bk, T + 1, n, 1), device=dev),
                   actions=torch.empty((Bk, T, m, 1), device=dev),
                   costs=torch.empty((Bk, T + 1, 1, 1), device=dev),
                   # every LQR kernel writes status[b] of every instance it is launched on: no memset kernel per call
                   status=(torch.zeros if B == 0 else torch.empty)((Bk,), dtype=torch.int32, device=dev))
        odt = torch.bfloat16 if storage_bf16 else torch.float32
        if storage_bf16 and self._suffix != "_f32":
            raise NotImplementedError("16-bit outputs are served for symmetric C (the fast kernels and the wave kernel)")
        if want_policy:
            out.update(K=torch.empty((Bk, T, m, n), device=dev, dtype=odt), k=torch.empty((Bk, T, m, 1), device=dev, dtype=odt))
        if want_value:
            out.update(V=torch.empty((Bk, T, n, n), device=dev, dtype=odt), v=torch.empty((Bk, T, n, 1), device=dev, dtype=odt),
                       const=torch.empty((Bk, T, 1, 1), device=dev, dtype=odt))
        ws_bytes = 0
        if not want_policy or storage_bf16:
            ws_bytes = int(lib.tfmpc_lqr_workspace_bytes(Bk, n, m, T))
            if workspace is None or workspace.numel() * workspace.element_size() < ws_bytes:
                workspace = torch.empty((max(ws_bytes, 4) + 3) // 4, dtype=torch.float32, device=dev)
            ws_bytes = workspace.numel() * workspace.element_size()
        entry = "tfmpc_lqr_solve_bf16out_f32" if storage_bf16 else "tfmpc_lqr_solve" + self._suffix
        rc = getattr(lib, entry)(Bk, n, m, T, *self._ptr_args(), _hip.ptr(x0),
                                     _hip.ptr(out["states"]), _hip.ptr(out["actions"]), _hip.ptr(out["costs"]),
                                     _hip.ptr(out.get("K")), _hip.ptr(out.get("k")), _hip.ptr(out.get("V")),
                                     _hip.ptr(out.get("v")), _hip.ptr(out.get("const")), _hip.ptr(out["status"]),
                                     _hip.ptr(workspace), ws_bytes, _hip.stream())
        _hip.check(rc, "tfmpc_lqr_solve_f32")
        self.last_status = out["status"]
        out["batched"] = B is not None
        out["workspace"] = workspace
        return out

    # -- lqr.py:163-166 ------------------------------------------------------------
    def solve(self, x0, T):
        out = self.solve_device(x0, T)
        states, actions, costs = out["states"], out["actions"], out["costs"]
        if not out["batched"]:
            states, actions, costs = states[0], actions[0], costs[0]
        return trajectory.Trajectory(states, actions, costs)

    # -- lqr.py:168-181 ------------------------------------------------------------
    def dump(self, file):
        config = {name: getattr(self, name).cpu().numpy().tolist() for name in ("F", "f", "C", "c")}
        json.dump(config, file)

    @classmethod
    def load(cls, file, device=None):
        config = json.load(file)
        return cls(**{key: np.array(val).astype("f") for key, val in config.items()}, device=device)
