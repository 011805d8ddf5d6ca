"""Free-running oracle solves with decision margins, for the decision-trace parity tests (tests/test_ilqr_trace_gpu.py).

``solve_with_margins`` is oracle/ilqr_ref.py's ``ILQRRef.solve`` (the reference's ilqr.py:214-283) unrolled so that every
pass through the body of ilqr.py:238-270 leaves a record -- the fields of the device trace
(``tfmpc_ilqr_solve_trace_f32``) -- together with the MARGIN of the pass: the smallest relative distance of any comparison
it took (g_norm / residual against atol, J_hat - J(alpha) against 0 for every step size tried) from flipping.  A device
that differs from this fp32 restatement only by rounding must take the same decisions wherever the margin is clear.

Importable without a GPU; ``run_many`` farms instances out to worker processes (spawned: they never touch the GPU)."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

ATOL_MARGIN = 1e-3     # relative margin under which an atol comparison counts as a near-tie
COST_MARGIN = 3e-5     # ... and a cost comparison |J_hat - J| / |J_hat| (an fp32 cost sum carries ~1e-6)


def make_env(kind, cfg, dtype):
    from oracle import envs_ref
    if kind == "navigation":
        return envs_ref.Navigation(cfg["goal"], cfg["deceleration"]["center"], cfg["deceleration"]["decay"], cfg["low"],
                                   cfg["high"], dtype=dtype)
    if kind == "hvac":
        return envs_ref.HVAC(**cfg, dtype=dtype)
    if kind == "reservoir":
        return envs_ref.Reservoir(**cfg, dtype=dtype)
    raise ValueError(kind)


def solve_with_margins(o, x0, T, u_init):
    """-> (records, states[T+1,n], actions[T,m], costs[T+1], iteration)."""
    dt = o.dtype
    mu, delta = 0.0, 1.0
    x_hat, u_hat, c_hat = o.start(x0, T, u_init=u_init)
    records = []
    alphas = np.geomspace(1.0, o.alpha_min, 11)
    iteration = 0
    attempts = 0
    for iteration in range(o.max_iterations):
        models = o.derivatives(x_hat, u_hat)
        converged = False
        while True:
            K, k, J_hat, dV1, dV2 = o._backward(T, u_hat, *models, mu, delta)
            g_norm = np.mean(np.max(np.abs(k) / (np.abs(u_hat) + dt(1.0)), axis=1), axis=0)[0]
            rec = dict(iteration=iteration, mu=float(mu), delta=float(delta), J_hat=float(J_hat), g_norm=float(g_norm),
                       alpha_index=None, accepted=None, residual=None, J=None,
                       margin=abs(float(g_norm) - o.atol) / o.atol / ATOL_MARGIN)
            records.append(rec)
            if g_norm < o.atol:
                converged = True
                break
            accept = False
            for step, alpha in enumerate(alphas):
                a = dt(alpha)
                x, u, c, J, residual = o.forward(x_hat, u_hat, K, k, a)
                delta_J = -a * (dV1 + a * dV2)
                dcost = J_hat - J
                z = dcost / delta_J if delta_J > 0 else np.sign(dcost)
                rec["margin"] = min(rec["margin"], abs(float(dcost)) / max(abs(float(J_hat)), 1e-12) / COST_MARGIN)
                if z >= o.c1:
                    accept = True
                    break
            rec.update(alpha_index=step, accepted=bool(accept), residual=float(residual), J=float(J))
            rec["margin"] = min(rec["margin"], abs(float(residual) - o.atol) / o.atol / ATOL_MARGIN)
            if residual < o.atol:
                converged = True
                x_hat, u_hat, c_hat = x, u, c
                break
            if accept:
                delta = min(1 / o.delta_0, delta / o.delta_0)
                mu = mu * delta * (mu * delta > o.mu_min)
                x_hat, u_hat, c_hat = x, u, c
                break
            delta = max(o.delta_0, delta * o.delta_0)
            mu = max(o.mu_min, mu * delta)
            attempts += 1
            if attempts >= 64 or not np.isfinite(mu) or mu > 1e30:
                return records, x_hat[..., 0], u_hat[..., 0], c_hat, iteration
        if converged:
            break
    return records, x_hat[..., 0], u_hat[..., 0], np.asarray(c_hat), iteration


def _job(args):
    kind, cfg, x0, u0, T, dtype_name, max_iterations = args
    import torch
    torch.set_num_threads(1)                 # one core per worker: the steps are tiny
    from oracle import ilqr_ref
    dtype = np.float32 if dtype_name == "float32" else np.float64
    o = ilqr_ref.ILQRRef(make_env(kind, cfg, dtype), dtype=dtype, max_iterations=max_iterations)
    recs, xs, us, cs, it = solve_with_margins(o, np.asarray(x0, dtype=dtype), T, np.asarray(u0, dtype=dtype))
    return recs, np.asarray(xs, dtype=np.float64), np.asarray(us, dtype=np.float64), np.asarray(cs, dtype=np.float64).reshape(-1), int(it)


def run_many(kind, cfg, x0, u0, T, dtype_name, max_iterations, workers=None):
    """x0[B,n,1], u0[B,T,m,1] -> list of _job results, computed in spawned worker processes."""
    import concurrent.futures
    import multiprocessing
    jobs = [(kind, cfg, x0[b], u0[b], T, dtype_name, max_iterations) for b in range(len(x0))]
    # at most FOUR workers: the restatement's envs import torch, which opens the GPU device in every process that loads
    # it, and a GPU box allows six processes on its card at once (this one included)
    workers = workers or max(1, min(len(jobs), (os.cpu_count() or 2) - 1, 4))
    if workers == 1:
        return [_job(j) for j in jobs]
    ctx = multiprocessing.get_context("spawn")
    with concurrent.futures.ProcessPoolExecutor(max_workers=workers, mp_context=ctx) as pool:
        return list(pool.map(_job, jobs, chunksize=max(1, len(jobs) // (4 * workers))))
