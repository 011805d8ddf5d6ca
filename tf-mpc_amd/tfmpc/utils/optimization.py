"""Projected-Newton box-constrained QP -- drop-in for the reference's
``tfmpc/utils/optimization.py`` (``projected_newton_qp`` :6-101, ``_get_qp_indices``
:121-127), executed by the in-wave box-QP of the HIP library (the same device routine
the bounded iLQR backward pass calls once per timestep)."""

import numpy as np
import torch

from tfmpc import _hip


def _f32(a, device):
    if isinstance(a, torch.Tensor):
        return a.detach().to(device=device, dtype=torch.float32)
    return torch.as_tensor(np.asarray(a, dtype=np.float32), device=device)


def projected_newton_qp(H, q, low, high, x, eps=1e-6, alpha_0=1.0, rho=0.5, c=1e-4):
    """``min 1/2 x^T H x + q^T x  s.t.  low <= x <= high`` from start point ``x``.

    ``H[m,m]`` (or ``[B,m,m]``), vectors ``[m,1]`` (or ``[B,m,1]``).  Returns
    ``(x, Hfree, free, clamped)`` like the reference: ``Hfree`` is the lower Cholesky
    factor of ``H`` restricted to the final free set (``None`` when batched)."""
    lib = _hip.require_gpu()
    dev = _hip.default_device()
    H = _f32(H, dev)
    batched = H.dim() == 3
    m = H.shape[-1]
    vec = lambda a: _f32(a, dev).reshape(-1, m).contiguous()
    Hb = H.reshape(-1, m, m).contiguous()
    B = Hb.shape[0]
    qv, lo, hi, x0 = (vec(a).expand(B, m).contiguous() for a in (q, low, high, x))
    xo = torch.empty((B, m), device=dev)
    free = torch.empty((B, m), device=dev)
    status = torch.zeros((B,), dtype=torch.int32, device=dev)
    rc = lib.tfmpc_boxqp_f32(B, m, _hip.ptr(Hb), _hip.ptr(qv), _hip.ptr(lo), _hip.ptr(hi), _hip.ptr(x0), _hip.ptr(xo),
                             _hip.ptr(free), _hip.ptr(status), _hip.stream())
    _hip.check(rc, "tfmpc_boxqp_f32")
    free_b = free != 0
    if batched:
        projected_newton_qp.last_status = status
        return xo.unsqueeze(-1), None, free_b.unsqueeze(-1), (~free_b).unsqueeze(-1)
    if int(status[0]) & _hip.ST_NOT_PD:
        raise ValueError("[boxQP] Hessian is not positive definite.")
    projected_newton_qp.last_status = status
    idx = torch.nonzero(free_b[0]).flatten()
    if int(status[0]) & _hip.ST_QP_LATER_NOT_PD:
        # a factorisation after the first failed: the reference breaks out (optimization.py:47-51) and returns the factor of the
        # PREVIOUS free set next to the free mask of the failing one; the stale factor is not reproduced here (None)
        return xo[0].unsqueeze(-1), None, free_b[0].unsqueeze(-1), (~free_b[0]).unsqueeze(-1)
    Hfree = torch.linalg.cholesky(Hb[0][idx][:, idx]) if idx.numel() else Hb[0][:0, :0]
    return xo[0].unsqueeze(-1), Hfree, free_b[0].unsqueeze(-1), (~free_b[0]).unsqueeze(-1)


def _get_qp_indices(g, low, high, x, eps=1e-6):
    """Clamped / free split (optimization.py:121-127); plain tensor logic."""
    g, low, high, x = (torch.as_tensor(np.asarray(a.cpu() if hasattr(a, "cpu") else a, dtype=np.float32))
                       for a in (g, low, high, x))
    c = ((x - low).abs() < eps) & (g > 0) | ((high - x).abs() < eps) & (g < 0)
    return ~c, c
