"""TEACHER-FORCED per-pass comparison of a fused whole-solve iLQR kernel with the restatement, at workload scale.

A free-running comparison of decision traces (tests/trace_oracle.py, tests/test_ilqr_lq_trace_gpu.py) ends at an instance's first
near-tie: from there on two fp32 programs may legitimately part ways, and on a workload whose passes are mostly near-ties (the
control-limited LQ problems: 17 % of the passes compared in round 4) it says little.  Here EVERY pass the device made is checked on
its own.  For pass p of instance b the device trace (``tfmpc_ilqr_solve_trace_f32``) holds the iteration index, mu, delta, the number
of local regularisation bumps the backward pass needed (``level``), J_hat, g_norm, the step size the line search ended on, its J and
residual and whether it was accepted; the nominal trajectory the device held when that pass started is recovered from launches
with ``max_iterations = k`` (the kernels are deterministic: run k extends run k - 1 bit for bit -- asserted by the caller).  The
restatement (oracle/ilqr_ref.py, restating /root/reference/tfmpc/solvers/ilqr.py:94-172,285-355,364-387 and
utils/optimization.py:6-101) is then handed the DEVICE's (x_hat, u_hat, mu, delta) and performs that ONE pass, in fp32 and in fp64:

* its own DECISIONS -- level (ilqr.py:305-309), g_norm < atol (:245), first accepted step size (:322-353), residual < atol (:253)
  -- with the margin of every comparison it took (pivots, clamp tests, atol tests, cost comparisons; constants of
  tests/trace_oracle.py).  Where the fp32 restatement's margin is clear and the fp64 restatement decides alike, the device must have
  decided the same; otherwise the pass is a near-tie for decisions (counted).
* its NUMBERS under the device's decisions -- backward pass at the device's level, rollout at the device's step size: J_hat,
  g_norm, J, residual and the candidate trajectory (which pins ``u = clip(u_hat + alpha k + K dx)``, hence K and k, at every step)
  must lie within 5 x the fp32 restatement's own error against fp64 (floors spelled out in `compare_pass`).  This part does
  not stop at ties: a near-tie decision taken either way still has to be followed by the right arithmetic.

Importable without a GPU; the passes are farmed out to spawned worker processes that never touch the GPU."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import trace_oracle  # noqa: E402

ALPHAS = np.geomspace(1.0, 1e-3, 11)


def _bumped(o, mu, delta, level):
    """mu after `level` local bumps of ilqr.py:308-309."""
    for _ in range(level):
        delta = max(o.delta_0, delta * o.delta_0)
        mu = max(o.mu_min, mu * delta)
    return mu


def one_pass(o, x_hat, u_hat, mu, delta, dev_level, dev_alpha_index, dev_k=None, dev_free=None):
    """One pass through the body of ilqr.py:238-270 from (x_hat[T+1,n,1], u_hat[T,m,1], mu, delta).

    -> dict(free=dict(level, converged_g, alpha_index, accepted, small_step, margin, later_failures[, k, selector_margin]) | None,
            forced=dict(J_hat, g_norm, J, residual, x, u, c) | None)
    `free`: the restatement's own decisions and the smallest margin of any comparison behind them (1 = the thresholds of
    tests/trace_oracle.py).  `forced`: its numbers with the backward pass at `dev_level` bumps and the rollout at step size
    `dev_alpha_index` (None: the device made no line search); None if the restatement cannot factorise at the device's level.

    `dev_k[T,m,1]` (bang-bang envs, ilqr.py:140-141 -- HVAC / Reservoir): the open-loop step the DEVICE's selector
    built (k_i = low - u or high - u by the sign of Q_u,i; read off the actions it moved).  The selector has exact ties on
    Reservoir in every sweep, so the line search and the rollout are restated on the device's direction (K = 0, dV1 = k_dev . Q_u,
    dV2 = 0) while `free["k"]` / `free["selector_margin"]` report the restatement's own selector entry by entry.

    `dev_free[T,m]` (bool; control-limited envs, round 6): the free set the DEVICE's box-QP ended on at every step of this pass
    (`clamp_mask` of tfmpc_ilqr_solve_trace_qp_f32).  The `forced` numbers are then computed with K_t on THAT set (ilqr.py:375-385: solve the
    free-set Newton system, K_clamped = 0; k_t stays the restatement's own QP solution) -- the discrete part of the backward pass is taken from
    the device, so the numbers can be compared without an excuse clause; `free["own_free"]` is the set the restatement's own QP ended on."""
    from oracle import ilqr_ref
    dt = o.dtype
    T = u_hat.shape[0]
    x_hat, u_hat = np.asarray(x_hat, dtype=dt), np.asarray(u_hat, dtype=dt)
    models = o.derivatives(x_hat, u_hat)
    gn = lambda k: float(np.mean(np.max(np.abs(k) / (np.abs(u_hat) + dt(1.0)), axis=1), axis=0)[0])
    bang = {}

    def on_bang_bang(t, Q_u, terms):
        bang[t] = (np.asarray(Q_u, dtype=np.float64).reshape(-1), np.asarray(terms, dtype=np.float64).reshape(-1))

    o.on_bang_bang = on_bang_bang
    o.forced_free, o.own_free = None, {}
    free = None
    with trace_oracle._PivotLog(value_tests=True) as pivots:
        try:
            level, mu_l, delta_l = 0, mu, delta
            while True:                                   # ilqr.py:285-315, as ILQRRef._backward, counting the local bumps
                bang.clear()
                try:
                    K, k, J_hat, dV1, dV2 = o.backward(T, u_hat, *models, dt(mu_l))
                    break
                except ilqr_ref.CholeskyFailure:
                    delta_l = max(o.delta_0, delta_l * o.delta_0)
                    mu_l = max(o.mu_min, mu_l * delta_l)
                    level += 1
                    if mu_l > 1e30:
                        raise
            pivot, failures, qp_sign = pivots.take()
            later_failures = failures - level             # factorisations that failed INSIDE a box-QP after its first (optimization.py:47-51)
            free = dict(level=int(level), alpha_index=None, accepted=None, small_step=None, later_failures=int(later_failures))
            if len(o.own_free) == T:
                free["own_free"] = np.stack([o.own_free[t] for t in range(T)])
            # discrete decisions INSIDE the backward pass: positive definite or not, which side of a bound a clamp test fell, and the
            # box-QP's own comparisons of objective values (Armijo, "no more improvement"): a pass whose smallest such margin is
            # under 1 can end on another free set -- other rows of K zero -- in another fp32 program
            margin = min(pivot / trace_oracle.PIVOT_MARGIN, qp_sign / trace_oracle.QP_MARGIN)
            free["internal_margin"] = float(min(margin, pivots.take_value_margin()))
            if len(bang) == T:                            # bang-bang at every step: entry-wise selector report
                Q_u = np.stack([bang[t][0] for t in range(T)])
                terms = np.stack([bang[t][1] for t in range(T)])
                free["k"] = np.asarray(k, dtype=np.float64)[..., 0]
                free["selector_margin"] = np.where(terms > 0, np.abs(Q_u) / np.where(terms > 0, terms, 1.0), np.inf)
                if dev_k is not None:
                    k = np.asarray(dev_k, dtype=dt).reshape(k.shape)
                    K = np.zeros_like(K)
                    dV1, dV2 = dt((np.asarray(k, dtype=np.float64)[..., 0] * Q_u).sum()), dt(0.0)
                else:
                    margin = min(margin, float(np.min(free["selector_margin"])) / (trace_oracle.SELECTOR_MARGIN * max(T, 1)))
            g_norm = gn(k)
            margin = min(margin, abs(g_norm - o.atol) / o.atol / trace_oracle.ATOL_MARGIN)
            free["converged_g"] = bool(g_norm < o.atol)
            if not free["converged_g"]:
                accept = False
                for step, alpha in enumerate(ALPHAS):
                    a = dt(alpha)
                    x, u, c, J, residual = o.forward(x_hat, u_hat, K, k, a)
                    delta_J = -a * (dV1 + a * dV2)
                    dcost = J_hat - J
                    z = dcost / delta_J if delta_J > 0 else np.sign(dcost)
                    if np.isfinite(float(dcost)):
                        margin = min(margin, abs(float(dcost)) / max(abs(float(J_hat)), 1e-12) / trace_oracle.COST_MARGIN)
                    else:
                        margin = 0.0
                    if z >= o.c1:
                        accept = True
                        break
                # the residual comparison: its distance from atol against the forward-error bound of an fp32 evaluation (trace_oracle)
                dx = np.abs(np.asarray(x[:-1], dtype=np.float64) - np.asarray(x_hat[:-1], dtype=np.float64))
                mag = np.abs(float(a) * np.asarray(k, dtype=np.float64)) + np.abs(np.asarray(K, dtype=np.float64)) @ dx
                noise = 8 * (K.shape[2] + 1) * 2.0 ** -24 * float(mag.max())
                margin = min(margin, abs(float(residual) - o.atol) / max(o.atol * trace_oracle.ATOL_MARGIN, noise))
                free.update(alpha_index=step, accepted=bool(accept), small_step=bool(residual < o.atol))
            free["margin"] = float(margin)
        except (ilqr_ref.CholeskyFailure, FloatingPointError):
            free = None
    forced = None
    o.own_free = None
    try:
        same_sets = dev_free is None or (free is not None and "own_free" in free and np.array_equal(free["own_free"], np.asarray(dev_free, dtype=bool)))
        if free is None or free["level"] != dev_level or not same_sets:
            o.forced_free = None if dev_free is None else np.asarray(dev_free, dtype=bool)
            K, k, J_hat, dV1, dV2 = o.backward(T, u_hat, *models, dt(_bumped(o, mu, delta, dev_level)))
            o.forced_free = None
        forced = dict(J_hat=float(J_hat), g_norm=gn(k), J=None, residual=None, x=None, u=None, c=None)
        if dev_alpha_index is not None:
            x, u, c, J, residual = o.forward(x_hat, u_hat, K, k, dt(ALPHAS[dev_alpha_index]))
            forced.update(J=float(J), residual=float(residual), x=np.asarray(x, dtype=np.float64)[..., 0],
                          u=np.asarray(u, dtype=np.float64)[..., 0], c=np.asarray(c, dtype=np.float64).reshape(-1))
    except ilqr_ref.CholeskyFailure:
        forced = None
    o.on_bang_bang = None
    o.forced_free = None
    return dict(free=free, forced=forced)


def _job(args):
    kind, cfg, x_hat, u_hat, mu, delta, dev_level, dev_alpha_index, dev_k, dtypes = args[:10]
    dev_free = args[10] if len(args) > 10 else None
    import torch
    torch.set_num_threads(1)
    from oracle import ilqr_ref
    out = {}
    for name in dtypes:
        dtype = np.float64 if name == "float64" else np.float32
        xh, uh, c = x_hat, u_hat, cfg
        if name == "float32p":               # fp32 on inputs moved by ~1 ulp: how much of each number is rounding noise on THIS pass
            rng = np.random.default_rng(4242)
            jig = lambda a: a if a is None or np.isscalar(a) else (np.asarray(a, dtype=np.float32) * (1 + 2.0 ** -23 * rng.uniform(-1, 1, size=np.shape(a)))).astype(np.float32)
            xh, uh = jig(x_hat), jig(u_hat)
            if kind == "lq":
                c = {k: jig(v) for k, v in cfg.items()}
        o = ilqr_ref.ILQRRef(trace_oracle.make_env(kind, c, dtype), dtype=dtype)
        with np.errstate(all="ignore"):
            out[name] = one_pass(o, xh, uh, mu, delta, dev_level, dev_alpha_index, dev_k, dev_free)
    return out


def run_passes(jobs, workers=None):
    """jobs: list of (kind, cfg, x_hat[T+1,n,1], u_hat[T,m,1], mu, delta, dev_level, dev_alpha_index, dev_k | None, dtype names[, dev_free[T,m] | None])
    -> list of dicts {dtype name: one_pass(...)}."""
    import concurrent.futures
    import multiprocessing
    # at most FOUR workers: `import torch` opens no GPU device, but the restatement's envs differentiate with torch.autograd, whose engine
    # asks every backend for its device count when it starts -- on ROCm that initialises the runtime and opens the card -- and a GPU
    # box allows six processes on its card at once, this one included (measured: twelve workers -> "13 processes had the GPU open")
    workers = workers or max(1, min(len(jobs), (os.cpu_count() or 2) - 1, 4))
    if workers == 1:
        return [_job(j) for j in jobs]
    ctx = multiprocessing.get_context("spawn")
    with concurrent.futures.ProcessPoolExecutor(max_workers=workers, mp_context=ctx) as pool:
        return list(pool.map(_job, jobs, chunksize=max(1, len(jobs) // (8 * workers))))


def sample_passes(n_passes, cap):
    """All passes of a short trace; of a long one the first 4, the last 2 and an even spread in between (`cap` in all)."""
    if n_passes <= cap:
        return list(range(n_passes))
    keep = set(range(4)) | {n_passes - 2, n_passes - 1}
    keep |= set(int(round(v)) for v in np.linspace(4, n_passes - 3, cap - len(keep)))
    return sorted(keep)


def compare_pass(d, nxt, r32, r64, r32p=None, atol=5e-3, traj_floor=2e-6, forced_sets=False):
    """One device pass `d` (a trace_records row) + the trajectory `nxt` = (x[T+1,n], u[T,m], c[T+1]) the device held afterwards (or
    None if the pass changed nothing) against the restatement's results of the same pass.

    -> (decision verdict, number verdict, detail):  decision: "same" | "tie" | "mismatch";  numbers: "ok" | "loose" | "mismatch" |
    "excused" (out of tolerance, but the pass holds a near-tie inside its backward pass) | "unposed" (the restatement cannot factorise
    at the device's level, or fp32 / fp64 overflow).  `forced_sets`: the restatement's numbers were computed on the DEVICE's box-QP free
    sets (one_pass(dev_free=...)): what a near-tie could change discretely has been taken from the device, so nothing is excused."""
    f32, f64 = r32["free"], r64["free"]
    dev_dec = dict(level=d["level"], converged_g=d["accepted"] is None, alpha_index=d["alpha_index"], accepted=d["accepted"],
                   small_step=None if d["residual"] is None else bool(d["residual"] < atol))
    keys = ("level", "converged_g", "alpha_index", "accepted", "small_step")
    decision, detail = "tie", ""
    others = [r64] + ([r32p] if r32p is not None else [])
    if f32 is not None and f32["margin"] >= 1.0 and all(o["free"] is not None and all(o["free"][k] == f32[k] for k in keys) for o in others):
        bad = [k for k in keys if dev_dec[k] != f32[k]]
        decision = "mismatch" if bad else "same"
        if bad and f32["internal_margin"] < 1.0:          # (a near-tie inside the backward pass, see below: the gains may differ discretely)
            decision = "tie"
        if bad:
            detail = f"decision {bad}: device {[dev_dec[k] for k in bad]}, restatement {[f32[k] for k in bad]} (margin {f32['margin']:.1f})"
    g32, g64 = r32["forced"], r64["forced"]
    if g32 is None or g64 is None:
        return decision, "unposed", detail
    worst, what = 0.0, ""
    # floors: two fp32 programs that sum ~50 costs in different orders differ by a few 1e-6 of the sum even when the restatement's own
    # error happens to be tiny; g_norm / residual are compared with atol, so they are held to 0.1 % of their size or of atol
    # (J sums the costs of the CANDIDATE trajectory, which is itself only pinned to `traj_floor` of its size -- on the control-limited workload
    # the box-QP's own stopping rule, optimization.py:27-29, leaves k_t that loose -- so J is held to the larger of the two relative bounds)
    quantities = [("J_hat", 2e-5, 1e-6), ("g_norm", 1e-3, 1e-3 * atol)] + ([("J", max(2e-5, traj_floor), 1e-6), ("residual", 1e-3, 1e-3 * atol)] if d["alpha_index"] is not None else [])
    for key, rtol, floor in quantities:
        ref = g64[key]
        if ref is None or not np.isfinite(ref) or not np.isfinite(g32[key]):
            return decision, "unposed", detail
        noise = max([abs(g32[key] - ref)] + ([abs(r32p["forced"][key] - ref)] if r32p is not None and r32p["forced"] is not None else []))
        tol = max(5 * noise, rtol * abs(ref), floor)
        ratio = abs(d[key] - ref) / tol
        if ratio > worst:
            worst, what = ratio, f"{key}: device {d[key]!r}, fp64 {ref!r}, fp32 {g32[key]!r}"
    if nxt is not None and g64["x"] is not None:
        for got, key in zip(nxt, ("x", "u", "c")):
            ref = g64[key]
            if not np.all(np.isfinite(ref)) or not np.all(np.isfinite(g32[key])):
                return decision, "unposed", detail
            scale = max(np.abs(ref).max(), 1.0)
            noise = max([np.abs(g32[key] - ref).max()] + ([np.abs(r32p["forced"][key] - ref).max()] if r32p is not None and r32p["forced"] is not None and r32p["forced"][key] is not None else []))
            tol = max(5 * noise, traj_floor * scale)
            ratio = np.abs(np.asarray(got, dtype=np.float64).reshape(ref.shape) - ref).max() / tol
            if ratio > worst:
                worst, what = ratio, f"candidate {key}: err {ratio * tol:.3e}, tolerance {tol:.3e} (scale {scale:.3e})"
    numbers = "ok" if worst <= 1.0 else ("loose" if worst <= 4.0 else "mismatch")
    # A pass with a near-tie INSIDE its backward pass (`internal_margin`: a clamp test, a pivot, one of the box-QP's value comparisons)
    # may legitimately have ended a box-QP on another free set: other rows of K_t are zero from that step on, and no tolerance on the
    # numbers downstream can hold.  Such a pass is "excused" -- counted, and capped in number by the caller.
    # (the fp32 runs' margins count: the fp64 run meets the same comparisons far from ITS rounding level)
    if not forced_sets and numbers != "ok" and (f32 is None or f32["internal_margin"] < 1.0 or (r32p is not None and (r32p["free"] is None or r32p["free"]["internal_margin"] < 1.0))):
        numbers = "excused"
    if numbers != "ok":
        detail = (detail + "; " if detail else "") + f"{what} ({worst:.1f} x tolerance)"
    return decision, numbers, detail
