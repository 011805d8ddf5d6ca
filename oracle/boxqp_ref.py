"""TEST INFRASTRUCTURE ONLY -- numpy restatement of
``tfmpc/utils/optimization.py:6-101,121-127`` (projected-Newton box QP, a port
of Tassa's ``boxQP``), single instance, column vectors ``[m, 1]``.
"""

import numpy as np


class BoxQPFactorizationError(Exception):
    """First Cholesky failed.  The reference leaves ``Hfree`` unbound there
    (optimization.py:47-51,101 -> UnboundLocalError, quirk Q11)."""


def get_qp_indices(g, low, high, x, eps=1e-6):
    """optimization.py:121-127"""
    clamped = np.logical_or(
        np.logical_and(np.abs(x - low) < eps, g > 0),
        np.logical_and(np.abs(high - x) < eps, g < 0))
    return np.logical_not(clamped), clamped


def projected_newton_qp(H, q, low, high, x, dtype=np.float64, eps=1e-6,
                        return_trace=False, monitor=None):
    """optimization.py:6-101.  Returns ``(x, Hfree, free, clamped)`` exactly as
    the reference does: ``free/clamped`` are those of the LAST index evaluation
    and ``Hfree`` the lower Cholesky factor of the last factorised ``H_ff``.

    ``monitor(kind, distance, scale)`` (test hook, tests/trace_oracle.py): called at the
    two comparisons of OBJECTIVE VALUES the loop takes -- "improvement" (:27-29: the
    distance of ``old_value - value`` from ``rtol |old_value|``) and "armijo" (:86: the
    distance of ``vc - old_value`` from ``armijo step sdotg``) -- with ``scale`` = the sum
    of the magnitudes of the terms of the objective at the point (what its rounding
    error is proportional to).  Changes nothing in the iteration."""
    H = np.asarray(H, dtype=dtype)
    q = np.asarray(q, dtype=dtype).reshape(-1, 1)
    low = np.asarray(low, dtype=dtype).reshape(-1, 1)
    high = np.asarray(high, dtype=dtype).reshape(-1, 1)
    x = np.asarray(x, dtype=dtype).reshape(-1, 1)

    def fobj(x):                                           # :8-11
        return (dtype(0.5) * (x.T @ (H @ x)) + q.T @ x).reshape(())

    def fscale(x):                                         # (monitor only)
        ax = np.abs(x).astype(np.float64)
        return float((0.5 * (ax.T @ (np.abs(H).astype(np.float64) @ ax)) + np.abs(q).astype(np.float64).T @ ax).reshape(()))

    max_iterations = 100                                   # :13-17
    rtol = dtype(1e-8)
    step_dec = dtype(0.6)
    min_step = dtype(1e-22)
    armijo = dtype(0.1)
    eps = dtype(eps)

    clamped = np.zeros(x.shape, dtype=bool)
    free = np.ones(x.shape, dtype=bool)
    Hfree = None
    value = fobj(x)
    old_value = value
    trace = []

    moved = True
    for iteration in range(max_iterations):                # :24
        if monitor is not None and iteration > 0 and moved:     # (an iterate that did not move has value == old_value in every program)
            monitor("improvement", float(old_value - value) - float(rtol * np.abs(old_value)), fscale(x))
        if iteration > 0 and (old_value - value) < rtol * np.abs(old_value):
            trace.append("improvement")
            break
        old_value = value
        old_clamped = clamped

        g = q + H @ x                                      # :34
        free, clamped = get_qp_indices(g, low, high, x, eps=dtype(1e-6))

        factorize = iteration == 0 or bool(np.any(old_clamped != clamped))
        if factorize:                                      # :40-51
            fr = free[:, 0]
            H_ff = H[np.ix_(fr, fr)]
            try:
                Hfree = np.linalg.cholesky(H_ff) if H_ff.size else H_ff
            except np.linalg.LinAlgError:
                if Hfree is None:
                    raise BoxQPFactorizationError()
                trace.append("not_pd")
                break

        if np.all(clamped):                                # :53-55
            trace.append("all_clamped")
            break

        fr = free[:, 0]
        grad_norm = np.sqrt(np.sum(g[fr] ** 2, dtype=dtype))   # :58
        if grad_norm < eps:
            trace.append("grad_norm")
            break

        grad_clamped = q + H @ (x * clamped.astype(dtype))     # :65
        search = np.zeros_like(x)
        rhs = grad_clamped[fr]
        y = np.linalg.solve(Hfree, rhs)                        # cholesky_solve
        sol = np.linalg.solve(Hfree.T, y)
        search[fr] = -sol - x[fr]                              # :70

        sdotg = (search.T @ g).reshape(())                     # :75
        if sdotg >= 0:
            trace.append("not_descent")
            break

        step = dtype(1.0)                                      # :82-95
        xc = np.clip(x + step * search, low, high)
        vc = fobj(xc)
        if monitor is not None:
            monitor("armijo", float(vc - old_value) - float(armijo * step * sdotg), fscale(xc))
        while (vc - old_value) / (step * sdotg) < armijo:
            step = step * step_dec
            xc = np.clip(x + step * search, low, high)
            vc = fobj(xc)
            if step < min_step:
                break
            if monitor is not None:
                monitor("armijo", float(vc - old_value) - float(armijo * step * sdotg), fscale(xc))

        moved = bool(np.any(xc != x))
        x = xc                                                 # :98-99
        value = vc
    else:
        trace.append("max_iterations")

    if return_trace:
        return x, Hfree, free, clamped, trace
    return x, Hfree, free, clamped
