"""DIRECT oracle tests of the cfg5 default kernel (``tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip``: sixteen instances per
wavefront, coupling products on the matrix cores) -- the kernel against the fp64 / fp32 restatement of
``/root/reference/tfmpc/solvers/ilqr.py:136-141,174-212,317-355`` and of the envs
(``envs/reservoir/__init__.py:47-79``, ``envs/hvac/__init__.py:69-129``), NOT against another kernel of the product
(the bit-identity tests in test_ilqr_costate_mfma_gpu.py stay, they are not a substitute).

What the fused solve exposes, and what is checked per instance (>= 16 different instances share one wave):

* start rollout (``ilqr.py:53-82``): a solve that "converges" at once returns it -- states and costs (their sum is
  the first sweep's J) against ``ILQRRef.start``;
* first backward sweep (``:136-141`` bang-bang branch): ``k_t = bound - u_t`` is rebuilt from the returned actions,
  ``u_new - u = alpha (bound - u)``, so the SELECTOR pattern (which bound each action is sent to = sign of Q_u) and
  the first accepted step size ``alpha`` (``:322-353``) are read off the result and compared with the oracle's
  wherever the oracle's own decision has a margin (|Q_u| above rounding; |J_hat - J(alpha)| above rounding);
* the state after one and after two iterations: decisions are discrete and flip between fp32 and fp64 (the fp32
  restatement itself lands elsewhere), so the arithmetic is checked decision by decision: the oracle's forward pass
  (``ILQRRef.forward``, fp64) is driven with the selector and step size the DEVICE chose, from the device's previous
  nominal trajectory, and must reproduce the device's states / actions / costs within the usual budget
  (5 x the fp32 restatement's own error).

Covered shapes: Reservoir and HVAC at n = 32, T = 100 (BASELINE configs[4]); the packed variants hvac6 (n = 6, two
instances per matrix-core column) and res4 (n = 4, four per column) from the reference's own config files."""

import numpy as np
import pytest
import torch

import problems
from oracle import envs_ref, ilqr_ref
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

pytestmark = pytest.mark.gpu

BUDGET = 5.0
ALPHAS = np.geomspace(1.0, 1e-3, 11)


MIN_DECIDED = 0.5          # share of the action entries that must have moved with a clear-cut selector (the fuzz lowers it)


def _case(kind, n, B):
    rng = np.random.default_rng(1000 + n)
    if kind == "hvac":
        cfg = dict(problems.HVAC6_CONFIG) if n == 6 else dict(problems.hvac_config(n, seed=5))
        x0 = rng.uniform(5.0, 30.0, size=(B, n, 1))
        return cfg, HVAC.load(dict(cfg)), envs_ref.HVAC, x0.astype(np.float32)
    cfg = dict(problems.RES4_CONFIG) if n == 4 else dict(problems.reservoir_config(n, seed=5))
    x0 = (np.array(problems.RES4_X0)[None] * rng.uniform(0.7, 1.3, size=(B, n, 1))) if n == 4 else rng.uniform(50.0, 75.0, size=(B, n, 1))
    return cfg, Reservoir.load(dict(cfg)), envs_ref.Reservoir, x0.astype(np.float32)


def _within(got, ref64, ref32, what):
    scale = np.abs(ref64).max()
    allowed = BUDGET * max(np.abs(ref32.astype(np.float64) - ref64).max(), 1e-6 * scale)
    err = np.abs(got.astype(np.float64) - ref64).max()
    assert err <= allowed, f"{what}: err {err:.3e} > allowed {allowed:.3e} (scale {scale:.3e})"


def _device(solver_kw, env, x0, T, u0):
    with _hip.option("TFMPC_ILQR_KERNEL", "costate_mfma"):
        out = iLQR(env, **solver_kw).solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items() if torch.is_tensor(v) and k != "workspace"}


def _oracle_sweep(o, xs, us):
    """One backward sweep of the fp64 oracle at (xs, us) plus the decision margins: relative size of Q_u against the
    magnitude of its own terms (K == 0, so V_x <- Q_x is the costate recursion)."""
    T = us.shape[0]
    tm, cm, fm = o.derivatives(xs, us)
    K, k, J, dV1, dV2 = o.backward(T, us, tm, cm, fm, mu=0.0)
    assert not K.any() and dV2 == 0.0                       # V_xx == 0, bang-bang branch every step
    V_x = fm.l_x
    margin, Q_u_all = np.empty_like(us), np.empty_like(us)
    for t in range(T - 1, -1, -1):
        Q_u = cm.l_u[t] + tm.f_u[t].T @ V_x
        mag = np.abs(cm.l_u[t]) + np.abs(tm.f_u[t]).T @ np.abs(V_x) + 1e-300
        margin[t], Q_u_all[t] = np.abs(Q_u) / mag, Q_u
        V_x = cm.l_x[t] + tm.f_x[t].T @ V_x
    assert np.isclose(dV1, (k * Q_u_all).sum(), rtol=1e-9)     # ilqr.py:166 with this recursion's Q_u
    return k, J, dV1, margin, Q_u_all


def _oracle_line_search(o, xs, us, J_hat, k, dV1):
    """ilqr.py:322-353 in fp64: the accepted step size and the smallest |J_hat - J(alpha)| / |J_hat| met on the way."""
    K0 = np.zeros((us.shape[0], us.shape[1], xs.shape[1]))
    worst = np.inf
    for a in ALPHAS:
        *_, J, _ = o.forward(xs, us, K0, k, a)
        delta_J = -a * dV1
        assert delta_J > 0                                   # k opposes the gradient: the expected change is a decrease
        worst = min(worst, abs(J_hat - J) / abs(J_hat))
        if (J_hat - J) / delta_J >= 0.0:
            return a, worst
    return None, worst


def _check_iteration(o64, o32, xs_dev, us_dev, out_next, tag):
    """Decision parity (with margins) and teacher-forced arithmetic parity for ONE iteration that took the device from
    the nominal (xs_dev, us_dev) to out_next = (states, actions, costs)."""
    T, m = us_dev.shape[0], us_dev.shape[1]
    xs, us = xs_dev.astype(np.float64), us_dev.astype(np.float64)
    k64, J64, dV1, margin, Q_u = _oracle_sweep(o64, xs, us)
    x1, u1, c1 = (a.astype(np.float64) for a in out_next)
    du = u1 - us
    moved = np.abs(du) > 0
    # selector the device applied: the bound each moved action heads to (ilqr.py:140-141)
    sel_dev_high = du > 0
    sel_64_high = k64 > 0
    decided = moved & (np.abs(k64) > 0) & (margin > 1e-3)
    assert decided.mean() > MIN_DECIDED, (tag, decided.mean())
    assert np.array_equal(sel_dev_high[decided], sel_64_high[decided]), \
        (tag, "selector", int((sel_dev_high[decided] != sel_64_high[decided]).sum()), int(decided.sum()))
    # step size the device accepted: du = alpha * (bound - u)
    k_dev = np.where(sel_dev_high, 1.0 - us, 0.0 - us) * moved
    big = moved & (np.abs(k_dev) > 0.05)
    ratio = du[big] / k_dev[big]
    a_dev = ALPHAS[np.argmin(np.abs(np.log(ALPHAS) - np.log(np.median(ratio))))]
    # du is the fp32 difference of two numbers of size <= 1: absolute rounding ~1e-7 against alpha * |k| >= alpha / 20
    assert np.abs(ratio / a_dev - 1.0).max() < 1e-4 + 4e-6 / a_dev, (tag, "alpha", a_dev, ratio.min(), ratio.max())
    # the line search of ilqr.py:322-353 in fp64 on the candidate direction the device built (its selector; equal to
    # the oracle's own except on near-tie entries): the first accepted step size must be the device's whenever the
    # oracle's accept / reject comparisons on the way were not themselves within rounding
    same_selector = np.array_equal(sel_dev_high[moved & (np.abs(k64) > 0)], sel_64_high[moved & (np.abs(k64) > 0)])
    a64, ls_margin = _oracle_line_search(o64, xs, us, J64, k_dev, (k_dev * Q_u).sum())
    if ls_margin > 1e-5:
        assert a64 is not None and np.isclose(a_dev, a64), (tag, "accepted alpha", a_dev, a64, ls_margin)
    # arithmetic, decision by decision: the oracle's forward pass with the device's selector and step size
    K0 = np.zeros((T, m, xs.shape[1]))
    xf, uf, cf, Jf, _ = o64.forward(xs, us, K0, k_dev, a_dev)
    x32, u32, c32, _, _ = o32.forward(xs_dev, us_dev, K0.astype(np.float32), k_dev.astype(np.float32), np.float32(a_dev))
    _within(u1, uf, u32, f"{tag}.actions")
    _within(x1, xf, x32, f"{tag}.states")
    _within(c1, cf, c32, f"{tag}.costs")
    return a_dev, same_selector


@pytest.mark.parametrize("kind,n,T,B", [("reservoir", 32, 100, 16), ("hvac", 32, 100, 16),
                                         ("hvac", 6, 40, 40), ("reservoir", 4, 40, 72)])
def test_costate_mfma_against_the_oracle(kind, n, T, B):
    cfg, env, OEnv, x0 = _case(kind, n, B)
    o64 = ilqr_ref.ILQRRef(OEnv(**cfg, dtype=np.float64), dtype=np.float64)
    o32 = ilqr_ref.ILQRRef(OEnv(**cfg, dtype=np.float32), dtype=np.float32)
    u0 = iLQR(env).random_actions(T, B, seed=n).cpu().numpy()            # [B, T, m, 1], one scalar uniform per step (Q1)
    start = _device(dict(max_iterations=1, atol=1e9), env, x0, T, u0)     # g_norm < atol at once: the start rollout
    one = _device(dict(max_iterations=1), env, x0, T, u0)
    two = _device(dict(max_iterations=2), env, x0, T, u0)
    assert not start["status"].any() and not one["status"].any() and not two["status"].any()
    assert (start["iterations"] == 0).all() and (one["iterations"] == 0).all()
    # every instance of the batch sits in some column (and, for the packed shapes, sub-column) of a wave: check a
    # spread of them -- all 16 of the single wave at n = 32
    picks = range(B) if n == 32 else sorted(set(np.linspace(0, B - 1, 12).astype(int)))
    alphas_seen, flips = [], 0
    for b in picks:
        # (1) start rollout and first sweep's J (= sum of the start costs)
        xs64, us64, cs64 = o64.start(x0[b], T, u_init=u0[b])
        xs32, _, cs32 = o32.start(x0[b], T, u_init=u0[b])
        _within(start["states"][b], xs64, xs32, f"{kind}{n}[{b}].start.states")
        _within(start["costs"][b], cs64, cs32, f"{kind}{n}[{b}].start.costs")
        assert np.array_equal(start["actions"][b], u0[b])
        # (2) + (3) first iteration from the device's own start rollout, then the second from its first
        a1, s1 = _check_iteration(o64, o32, start["states"][b], start["actions"][b],
                                  (one["states"][b], one["actions"][b], one["costs"][b]), f"{kind}{n}[{b}].it1")
        if two["iterations"][b] == 1:                         # (not converged inside iteration 2's tests)
            a2, s2 = _check_iteration(o64, o32, one["states"][b], one["actions"][b],
                                      (two["states"][b], two["actions"][b], two["costs"][b]), f"{kind}{n}[{b}].it2")
            alphas_seen.append(a2)
            flips += (not s2)
        alphas_seen.append(a1)
        flips += (not s1)
        # the accepted candidate is an improvement (ilqr.py:339-353 with c1 = 0)
        assert one["costs"][b].sum() <= start["costs"][b].sum() * (1 + 1e-6)
    print(f"\n{kind} n={n}: step sizes accepted {sorted(set(np.round(alphas_seen, 4)))}, "
          f"{flips} of {len(alphas_seen)} sweeps had a near-tie selector entry that fp32 resolved differently")
