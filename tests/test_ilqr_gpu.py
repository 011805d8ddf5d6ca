"""GPU parity tests of the iLQR path (``-m gpu``): env kernels, derivatives, backward
(three controllers), forward, box-QP and the fused whole-solve kernel, each against the
fp64 oracle's committed golden vectors (PARITY UNPINNED for numeric iLQR outputs: the
golden vectors come from ``oracle/``, see ``tests/golden/make_golden.py``) and, for the
self-consistency properties, in the style of the reference's ``tests/test_ilqr.py``.

Tolerances: fp32 device arithmetic vs fp64 golden values, relative to each tensor's
max-abs: 2e-5 for single kernels fed golden inputs (one pass of fp32 rounding on
well-conditioned data; BASELINE.json asks for 1e-5, which holds on the navigation envs
and is asserted there), 1e-3 for whole solves (tens of chained iterations).
"""

import numpy as np
import pytest
import torch

import problems
from oracle import boxqp_ref, envs_ref, ilqr_ref
from tfmpc import _hip
from tfmpc.envs import make_env
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.lq import LQEnv
from tfmpc.envs.lqr.navigation import NavigationLQR
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
from tfmpc.utils import optimization

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _close(got, ref, rtol, what, atol=0.0):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30) if ref.size else 1.0
    err = np.abs(got - ref).max() if ref.size else 0.0
    assert err <= rtol * scale + atol, f"{what}: err {err:.3e} > {rtol:.0e} * {scale:.3e}"


def _cases():
    """(name, env factory, golden file, suffix) for every golden iLQR record."""
    out = []
    for i, (beta, bounds) in enumerate([(0.0, None), (0.0, (-1.0, 1.0)), (5.0, None), (5.0, (-1.0, 1.0))]):
        low, high = bounds if bounds else (None, None)
        out.append((f"navlqr{i}", lambda beta=beta, low=low, high=high: NavigationLQR([[5.5], [-9.0]], beta, low, high),
                    "ilqr_navlqr", str(i)))
    for i in range(3):
        out.append((f"nav{i}", lambda: Navigation.load(problems.NAV_CONFIG), "ilqr_navigation", str(i)))
    out.append(("hvac6", lambda: HVAC.load(dict(problems.HVAC6_CONFIG)), "ilqr_hvac6", ""))
    out.append(("res4", lambda: Reservoir.load(dict(problems.RES4_CONFIG)), "ilqr_res4", ""))
    out.append(("lq16x8", None, "ilqr_lq16x8", ""))
    return out


CASES = _cases()


def _env(case, g):
    name, factory, _, _ = case
    if factory is None:
        return LQEnv(g["lq_F"], g["lq_f"], g["lq_C"], g["lq_c"])
    return factory()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_start_and_derivatives_match_golden(golden, case):
    g = golden(case[2])
    sfx = case[3]
    G = lambda k: g[k + sfx]
    env = _env(case, g)
    solver = iLQR(env)
    T = int(G("T"))
    tight = 1e-5 if case[0].startswith("nav") else 2e-5
    xs, us, cs = solver.start(G("x0")[:, None], T, u_init=G("u_init")[..., None])
    assert xs.shape == (T + 1, env.state_size, 1) and us.shape == (T, env.action_size, 1) and cs.shape == (T + 1,)
    _close(_np(xs)[..., 0], G("start_states"), tight, "start.states")
    _close(_np(cs), G("start_costs"), tight, "start.costs")
    # feed the GOLDEN nominal trajectory so every derivative is evaluated at the same point
    tm, cm, fm = solver.derivatives(G("start_states")[..., None], G("u_init")[..., None])
    assert all(a.shape[0] == T for a in tm) and all(a.shape[0] == T for a in cm)      # reference tests/test_ilqr.py:65-74
    for key, got in (("f", tm.f[..., 0]), ("f_x", tm.f_x), ("f_u", tm.f_u), ("l", cm.l), ("l_x", cm.l_x[..., 0]),
                     ("l_u", cm.l_u[..., 0]), ("l_xx", cm.l_xx), ("l_uu", cm.l_uu), ("l_ux", cm.l_ux), ("l_xu", cm.l_xu),
                     ("fl", fm.l), ("fl_x", fm.l_x[..., 0]), ("fl_xx", fm.l_xx)):
        _close(_np(got), G(key), tight, f"{case[0]}.{key}", atol=1e-30)
    # DiffEnv single-call protocol (diffenv.py:13-101) agrees with the batched solver call
    lt = env.get_linear_transition(G("start_states")[:-1, :, None], G("u_init")[..., None], batch=True)
    assert torch.equal(lt.f_x, tm.f_x) and torch.equal(lt.f, tm.f)
    one = env.get_quadratic_cost(G("start_states")[0][:, None], G("u_init")[0][:, None], batch=False)
    assert torch.equal(one.l_x, cm.l_x[0]) and one.l.shape == ()
    assert torch.equal(env.transition(G("start_states")[0][:, None], G("u_init")[0][:, None]), xs[1])
    assert float(env.final_cost(G("start_states")[-1][:, None])) == pytest.approx(float(fm.l), rel=1e-6)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_backward_and_forward_match_golden(golden, case):
    g = golden(case[2])
    sfx = case[3]
    G = lambda k: g[k + sfx]
    env = _env(case, g)
    solver = iLQR(env)
    T = int(G("T"))
    from tfmpc.envs.diffenv import CostApprox, FinalCostApprox, TransitionApprox
    tm = TransitionApprox(G("f")[..., None], G("f_x"), G("f_u"))
    cm = CostApprox(G("l"), G("l_x")[..., None], G("l_u")[..., None], G("l_xx"), G("l_uu"), G("l_ux"), G("l_xu"))
    fm = FinalCostApprox(G("fl"), G("fl_x")[:, None], G("fl_xx"))
    u = G("u_init")[..., None]
    for tag, mu in (("mu0", 0.0), ("mu1", 1.0)):
        K, k, J, dV1, dV2 = solver.backward(T, u, tm, cm, fm, mu=mu)
        assert K.shape == (T, env.action_size, env.state_size) and k.shape == (T, env.action_size, 1)   # test_ilqr.py:77-89
        rt = 2e-4 if case[0] == "lq16x8" else 5e-5      # lq16x8: make_lqr conditioning
        if env.action_space.is_bounded() and np.any(G("K_mu0")):
            # gains come out of the box-QP, whose stopping rule is "objective improved by less
            # than 1e-8 relative" (optimization.py:14,27): in fp32 that resolves the minimiser
            # only to ~sqrt(eps_fp32) ~ 3e-4 of its scale
            rt = 1e-3
        _close(_np(K), G(f"K_{tag}"), rt, f"{case[0]}.K_{tag}", atol=1e-6)
        _close(_np(k)[..., 0], G(f"k_{tag}"), rt, f"{case[0]}.k_{tag}", atol=1e-6)
        _close(_np(J), G(f"J_{tag}"), 1e-5, f"{case[0]}.J_{tag}")
        scale = max(abs(float(G(f"dV1_{tag}"))), abs(float(G(f"dV2_{tag}"))), 1e-12)
        assert abs(float(dV1) - float(G(f"dV1_{tag}"))) <= rt * scale + 1e-6
        assert abs(float(dV2) - float(G(f"dV2_{tag}"))) <= rt * scale + 1e-6
    # default mu of backward is 1.0 (quirk Q5)
    K1 = solver.backward(T, u, tm, cm, fm)[0]
    assert torch.equal(K1, K)
    # forward with the GOLDEN gains
    xs = G("start_states")[..., None]
    for a in (0, 1):
        alpha = float(G(f"fwd{a}_alpha"))
        st, ac, co, Jn, res = solver.forward(xs, u, G("K_mu0"), G("k_mu0")[..., None], alpha)
        assert st.shape == xs.shape and ac.shape == u.shape and co.shape == (T + 1,)
        rt = 2e-4 if case[0] == "lq16x8" else 2e-5
        _close(_np(st)[..., 0], G(f"fwd{a}_states"), rt, f"{case[0]}.fwd{a}.states")
        _close(_np(ac)[..., 0], G(f"fwd{a}_actions"), rt, f"{case[0]}.fwd{a}.actions", atol=1e-6)
        _close(_np(co), G(f"fwd{a}_costs"), rt, f"{case[0]}.fwd{a}.costs")
        assert abs(float(Jn) - float(G(f"fwd{a}_J"))) <= rt * abs(float(G(f"fwd{a}_J")))
        assert abs(float(res) - float(G(f"fwd{a}_residual"))) <= rt * max(1.0, float(G(f"fwd{a}_residual")))
        # self-consistency as in the reference's tests/test_ilqr.py:92-109: the rollout obeys the env
        for t in (0, T // 2, T - 1):
            assert torch.equal(st[t + 1], env.transition(st[t], ac[t]))
            assert float(co[t]) == pytest.approx(float(env.cost(st[t], ac[t])), rel=1e-6, abs=1e-6)
        assert float(co[T]) == pytest.approx(float(env.final_cost(st[T])), rel=1e-6, abs=1e-6)


def _host_driven_solve(solver, x0, T, u_init):
    """The reference's solve loop (ilqr.py:214-283) driven from Python over the
    INDIVIDUAL GPU kernels; must reproduce the fused kernel bit for bit."""
    mu, delta = 0.0, 1.0
    x_hat, u_hat, c_hat = solver.start(x0, T, u_init=u_init)
    iteration = 0
    for iteration in range(solver.max_iterations):
        models = solver.derivatives(x_hat, u_hat)
        converged = False
        while True:
            K, k, J_hat, dV1, dV2 = solver.backward(T, u_hat, *models, mu=mu)
            g_norm = float(torch.mean(torch.amax(k.abs() / (u_hat.abs() + 1.0), dim=1), dim=0)[0])
            if g_norm < solver.atol:
                converged = True
                break
            accept = False
            for alpha in solver._alphas():
                a32 = np.float32(alpha)
                x, u, c, J, residual = solver.forward(x_hat, u_hat, K, k, float(a32))
                delta_J = -a32 * (np.float32(dV1.item()) + a32 * np.float32(dV2.item()))
                dcost = np.float32(J_hat.item()) - np.float32(J.item())
                z = dcost / delta_J if delta_J > 0 else np.sign(dcost)
                if z >= solver.c1:
                    accept = True
                    break
            if float(residual) < solver.atol:
                converged = True
                x_hat, u_hat, c_hat = x, u, c
                break
            if accept:
                delta = min(1 / solver.delta_0, delta / solver.delta_0)
                mu = mu * delta * (mu * delta > solver.mu_min)
                x_hat, u_hat, c_hat = x, u, c
                break
            delta = max(solver.delta_0, delta * solver.delta_0)
            mu = max(solver.mu_min, mu * delta)
        if converged:
            break
    return x_hat, u_hat, c_hat, iteration


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fused_solve_matches_golden_and_host_driven_loop(golden, case):
    g = golden(case[2])
    sfx = case[3]
    G = lambda k: g[k + sfx]
    env = _env(case, g)
    solver = iLQR(env)
    T = int(G("T"))
    x0, u0 = G("x0")[:, None], G("u_init")[..., None]
    # (a) the fused WAVE kernel == the reference's loop driven over the single kernels, bit for bit
    # (the LQ env would otherwise take the matrix-core solve kernel, which rounds differently)
    with _hip.option("TFMPC_ILQR_KERNEL", "wave"):
        traj, iteration = solver.solve(x0, T, show_progress=False, u_init=u0)
    assert int(solver.last_status[0]) & ~_hip.ST_QP_MAXITER == 0
    xh, uh, ch, it_h = _host_driven_solve(solver, x0, T, u0)
    assert iteration == it_h
    assert np.array_equal(traj.states, xh[..., 0].cpu().numpy()) and np.array_equal(traj.costs, ch.cpu().numpy())
    # default dispatch (matrix cores for lq16x8) is what (b) checks against the oracle
    traj, iteration = solver.solve(x0, T, show_progress=False, u_init=u0)
    assert int(solver.last_status[0]) & ~_hip.ST_QP_MAXITER == 0
    # (b) vs the oracle.  On the smooth LQ-type problems: same iteration count and the same
    # trajectory to fp32 accuracy.  On the nonlinear / piecewise-linear envs the 11-point line
    # search takes DISCRETE decisions that flip between fp32 and fp64 (the fp32 CPU restatement
    # itself lands 1.8 % away from the fp64 one on res4: 227.72 vs 223.77 after 27 vs 32
    # iterations), so there the converged cost is compared with BOTH restatements.
    if case[0] in ("hvac6", "res4") or (case[0].startswith("nav") and not case[0].startswith("navlqr")):
        oenv = {"hvac6": lambda: envs_ref.HVAC(**problems.HVAC6_CONFIG, dtype=np.float32),
                "res4": lambda: envs_ref.Reservoir(**problems.RES4_CONFIG, dtype=np.float32)}.get(
            case[0], lambda: envs_ref.Navigation(problems.NAV_CONFIG["goal"], problems.NAV_CONFIG["deceleration"]["center"],
                                                 problems.NAV_CONFIG["deceleration"]["decay"], problems.NAV_CONFIG["low"],
                                                 problems.NAV_CONFIG["high"], dtype=np.float32))()
        # Round 4: a decision-trace check instead of round 1's "within 1 % / 3 % of the cost, iterations within a third".  The device
        # trace (iLQR.solve(trace=True)) must equal the free-running fp32 restatement's pass by pass up to the restatement's first
        # near-tie (tests/trace_oracle.py: atol / cost comparisons and the bang-bang selector); with no near-tie at all the
        # iteration count is the restatement's and the cost agrees to fp32 accuracy, otherwise either side is right from the
        # tie on and the two converged costs may differ by what the restatements differ by among themselves (res4: 1.8 %).
        import trace_oracle
        from test_ilqr_trace_gpu import _compare
        from tfmpc.solvers.ilqr import trace_records
        o32 = ilqr_ref.ILQRRef(oenv, dtype=np.float32)
        records, _, _, c32, it32 = trace_oracle.solve_with_margins(o32, x0.astype(np.float32), T, u0.astype(np.float32))
        traj_t, it_t = solver.solve(x0, T, show_progress=False, u_init=u0, trace=True)
        assert it_t == iteration and np.array_equal(traj_t.costs, traj.costs)
        agreed, verdict = _compare(solver.last_trace[0], records)
        assert not verdict.startswith("mismatch"), verdict
        total64 = G("sol_costs").sum()
        if verdict == "full":
            assert iteration == it32 and abs(traj.total_cost - c32.sum()) <= 1e-4 * abs(c32.sum())
        else:
            if case[0] != "res4":                                      # (Reservoir: a selector operand cancels exactly in every first pass,
                assert agreed >= 1, (agreed, verdict)                  #  tests/test_ilqr_trace_gpu.py) -- elsewhere the first pass is tie-free
            assert abs(traj.total_cost - c32.sum()) <= 3e-2 * abs(c32.sum()) and abs(traj.total_cost - total64) <= 3e-2 * abs(total64)
    else:
        assert iteration == int(G("sol_iteration"))
        _close(traj.states, G("sol_states"), 1e-3, f"{case[0]}.sol.states")
        _close(traj.actions, G("sol_actions"), 1e-3, f"{case[0]}.sol.actions", atol=1e-5)
        _close(traj.costs, G("sol_costs"), 1e-3, f"{case[0]}.sol.costs")


def test_fused_solve_tracks_the_fp32_restatement_on_navigation():
    """Line-search decisions of the fp32 device path vs the fp32 CPU restatement."""
    cfg = problems.NAV_CONFIG
    env = Navigation.load(cfg)
    oenv = envs_ref.Navigation(cfg["goal"], cfg["deceleration"]["center"], cfg["deceleration"]["decay"], cfg["low"],
                               cfg["high"], dtype=np.float32)
    rng = np.random.default_rng(77)
    B, T = 16, 20
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = np.stack([problems.scalar_uniform_actions(T, [-1, -1], [1, 1], rng) for _ in range(B)]).astype(np.float32)
    traj, its = iLQR(env).solve(x0, T, u_init=u0)
    import trace_oracle
    clear = 0
    for b in range(B):
        o = ilqr_ref.ILQRRef(oenv, dtype=np.float32)
        records, x, u, c, it = trace_oracle.solve_with_margins(o, x0[b], T, u0[b])
        if min(r["margin"] for r in records) >= 1.0:
            # every decision of the restatement has a clear margin (tests/trace_oracle.py): the device takes the same
            # ones -- same iteration count, same cost to fp32 accuracy (round 3; the whole traces are compared pass by
            # pass on 256 instances in tests/test_ilqr_trace_gpu.py)
            clear += 1
            assert it == its[b], (b, it, its[b])
            assert abs(traj.total_cost[b] - c.sum()) <= 1e-4 * abs(c.sum()), b
        else:
            assert abs(traj.total_cost[b] - c.sum()) <= 2e-2 * abs(c.sum()), b      # a near-tie somewhere: either side is right
    assert clear >= (3 * B) // 4, clear


def test_batched_solve_equals_single_solves_and_shapes():
    env = NavigationLQR([[5.5], [-9.0]], 5.0, -1.0, 1.0)
    solver = iLQR(env)
    B, T = 9, 10
    rng = np.random.default_rng(0)
    x0 = rng.normal(size=(B, 2, 1)).astype(np.float32)
    u0 = solver.random_actions(T, B, seed=5)
    assert torch.equal(u0[..., 0, :], u0[..., 1, :])             # one scalar per step (quirk Q1)
    assert float(u0.min()) >= -1.0 and float(u0.max()) <= 1.0
    traj, its = solver.solve(x0, T, u_init=u0)
    assert traj.states.shape == (B, T + 1, 2) and traj.actions.shape == (B, T, 2) and its.shape == (B,)
    for b in (0, 4, 8):
        one, it = solver.solve(x0[b], T, u_init=u0[b])
        assert it == its[b] and np.array_equal(one.states, traj.states[b])
    # per-instance goals: one launch, B different problems
    goals = rng.uniform(-10, 10, size=(B, 2, 1)).astype(np.float32)
    benv = NavigationLQR(goals, 5.0)
    tb, itb = iLQR(benv).solve(x0, T, u_init=u0)
    for b in (1, 7):
        one, it = iLQR(NavigationLQR(goals[b], 5.0)).solve(x0[b], T, u_init=u0[b])
        assert it == itb[b] and np.array_equal(one.states, tb.states[b])


def test_unbounded_ilqr_reproduces_lqr_solution():
    """On an LQ env iLQR must land on the LQR optimum (one Newton step); checked on the
    BASELINE headline shape n=16, m=8 against the LQR kernel path."""
    from tfmpc.solvers.lqr import LQR
    F, f, C, c = problems.make_lqr_instance(1001, 16, 8)
    T = 12
    x0 = np.random.default_rng(1).normal(size=(16, 1)).astype(np.float32)
    lq = LQR(F, f, C, c).solve(x0, T)
    traj, it = iLQR(LQEnv(F, f, C, c)).solve(x0, T, u_init=np.zeros((T, 8, 1), dtype=np.float32))
    assert it <= 2
    assert np.abs(traj.states - lq.states).max() <= 2e-3 * np.abs(lq.states).max()
    assert abs(traj.total_cost - lq.total_cost) <= 2e-3 * np.abs(lq.costs).sum()


# ---------------------------------------------------------------------- box-QP ---
@pytest.mark.parametrize("case", range(6))
def test_boxqp_known_answers(golden, case):
    """The reference's tests/test_utils_optimization.py:57-79 against the HIP box-QP."""
    g = golden("boxqp_kats")
    goal, low, high, x_star = (g[f"{k}{case}"] for k in ("goal", "low", "high", "x_star"))
    m = len(goal)
    H, q = 2 * np.eye(m), -2 * goal[:, None]
    x, Hfree, free, clamped = optimization.projected_newton_qp(H, q, low[:, None], high[:, None], x_star[:, None])
    assert x.shape == (m, 1) and np.all(np.abs(_np(x)[:, 0] - x_star) < 1e-4)
    assert Hfree.shape[0] == int(free.sum()) and bool((free ^ clamped).all())
    rng = np.random.default_rng(case)
    starts = []
    for _ in range(10):
        x_0 = rng.uniform(low, high)
        starts.append(x_0)
        for bound in (low, high):
            for i in range(m):
                xs = x_0.copy()
                xs[i] = bound[i]
                starts.append(xs)
    starts = np.stack(starts)
    B = len(starts)
    xb, _, fb, cb = optimization.projected_newton_qp(np.broadcast_to(H, (B, m, m)), np.broadcast_to(q, (B, m, 1)),
                                                     np.broadcast_to(low[:, None], (B, m, 1)),
                                                     np.broadcast_to(high[:, None], (B, m, 1)), starts[..., None])
    assert np.all(np.abs(_np(xb)[..., 0] - x_star) < 1e-4)
    # interior points are all free (tests/test_utils_optimization.py:30-41)
    xin = rng.uniform(low + 1e-4, high - 1e-4)[:, None]
    fr, cl = optimization._get_qp_indices(q + H @ xin, low[:, None], high[:, None], xin)
    assert bool(fr.all()) and not bool(cl.any())


def test_boxqp_dense_matches_oracle(golden):
    g = golden("boxqp_dense")
    for i in range(int(g["n_cases"])):
        H, q, low, high, x0, x_ref, free_ref = (g[f"{k}{i}"] for k in ("H", "q", "low", "high", "x0", "x", "free"))
        x, _, free, _ = optimization.projected_newton_qp(H, q[:, None], low[:, None], high[:, None], x0[:, None])
        assert np.abs(_np(x)[:, 0] - x_ref).max() < 2e-4, i
        assert np.array_equal(free[:, 0].cpu().numpy(), free_ref), i


def test_boxqp_later_factorisation_failure_breaks_out_like_the_reference():
    """optimization.py:47-51: when a factorisation AFTER the first fails (the free set grew onto an indefinite block), the reference logs
    and breaks out of the loop -- x and the free mask are those at the break.  Constructed: H indefinite on coordinates (0, 1),
    coordinate 1 starts clamped at its upper bound (first free set {0, 2}: positive definite); the Newton step on coordinate 0 turns the
    gradient of coordinate 1 inward, the free set becomes everything, H itself does not factorise.  The restatement and the device return
    the same point and mask; the device says so in its status (TFMPC_ST_QP_LATER_NOT_PD).  From iLQR the case cannot arise in exact
    arithmetic: the QP starts at the box centre (ilqr.py:369), where the first factorisation is of the whole H."""
    H = np.array([[1.0, 2.0, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 2.0]])
    q = np.array([[-2.5], [-1.5], [-1.0]])
    low, high, x0 = -np.ones((3, 1)), np.ones((3, 1)), np.array([[0.0], [1.0], [0.0]])
    for dtype in (np.float32, np.float64):
        xr, Hf, fr, cl, tr = boxqp_ref.projected_newton_qp(H, q, low, high, x0, dtype=dtype, return_trace=True)
        assert tr == ["not_pd"] and Hf.shape == (2, 2) and fr.all() and np.allclose(xr[:, 0], [0.5, 1.0, 0.5])
    x, Hfree, free, clamped = optimization.projected_newton_qp(H, q, low, high, x0)
    assert int(optimization.projected_newton_qp.last_status[0]) == _hip.ST_QP_LATER_NOT_PD
    assert np.allclose(_np(x)[:, 0], [0.5, 1.0, 0.5], atol=1e-6) and bool(free.all()) and not bool(clamped.any()) and Hfree is None
    # batched, next to a problem that is fine: one flag, the other instance untouched
    Hb = np.stack([H, 2 * np.eye(3)])
    xb, _, fb, _ = optimization.projected_newton_qp(Hb, np.stack([q, q]), np.stack([low, low]), np.stack([high, high]), np.stack([x0, x0]))
    assert optimization.projected_newton_qp.last_status.tolist() == [_hip.ST_QP_LATER_NOT_PD, 0]
    assert np.allclose(_np(xb)[1, :, 0], [1.0, 0.75, 0.5], atol=1e-5)
    # the first factorisation failing is the other case: an exception (single instance), as the reference's UnboundLocalError -> exit
    with pytest.raises(ValueError):
        optimization.projected_newton_qp(H, q, low, high, np.zeros((3, 1)))


def test_non_pd_quu_raises_like_the_reference():
    """ilqr.py:305: a non-PD Q_uu_reg surfaces as an exception from backward (single
    instance) / a status bit (batch); solve() retries with more regularisation."""
    from tfmpc.envs.diffenv import CostApprox, FinalCostApprox, TransitionApprox
    env = NavigationLQR([[1.0], [1.0]], 1.0)
    solver = iLQR(env)
    T, n = 1, 2
    I = np.broadcast_to(np.eye(n), (T, n, n))
    tm = TransitionApprox(np.zeros((T, n, 1)), I, I)
    cm = CostApprox(np.zeros(T), np.zeros((T, n, 1)), np.zeros((T, n, 1)), 0 * I, -0.5 * I, 0 * I, 0 * I)
    fm = FinalCostApprox(np.zeros(()), np.zeros((n, 1)), np.zeros((n, n)))
    with pytest.raises(ArithmeticError):
        solver.backward(T, np.zeros((T, n, 1)), tm, cm, fm, mu=0.0)
    K, k, *_ = solver.backward(T, np.zeros((T, n, 1)), tm, cm, fm, mu=1.0)     # Q_uu_reg = 0.5 I: fine
    assert np.allclose(_np(K)[0], -2.0 * np.eye(n))
    # batched: a status bit per instance instead of an exception
    stack = lambda tup: type(tup)(*[np.stack([np.asarray(a)] * 2) for a in tup])
    solver.backward(T, np.zeros((2, T, n, 1)), stack(tm), stack(cm), stack(fm), mu=np.array([0.0, 1.0]))
    assert solver.last_status.tolist() == [_hip.ST_NOT_PD, 0]


def test_full_size_cfg4_navigation_batch_properties():
    """BASELINE.json cfg4 at full size: Navigation (nav.config.json), T=50, B=16 384,
    x0 ~ U(0,10)^2.  Size-independent properties on every instance."""
    env = make_env({"module": "navigation", "cls_name": "Navigation", "config": problems.NAV_CONFIG})
    solver = iLQR(env)
    B, T = 16384, 50
    rng = np.random.default_rng(4)
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = solver.random_actions(T, B, seed=4)
    out = solver.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    states, actions, costs = out["states"][..., 0], out["actions"][..., 0], out["costs"]
    assert torch.isfinite(states).all() and torch.isfinite(costs).all()
    assert int((out["status"] & _hip.ST_NAN).sum()) == 0
    assert float(actions.abs().max()) <= 1.0                                     # box respected
    start_cost = solver.start(x0, T, u_init=u0)[2].sum(dim=1)
    assert bool((costs.sum(dim=1) <= start_cost * (1 + 1e-5)).all())            # never worse than the start
    # the returned trajectory obeys the env: x_{t+1} == transition(x_t, u_t), cost_t == cost(x_t, u_t)
    idx = torch.arange(0, B, 64, device=states.device)
    for t in (0, 17, 49):
        nxt = env.transition(states[idx, t].unsqueeze(-1), actions[idx, t].unsqueeze(-1), batch=True)
        assert torch.equal(nxt[..., 0], states[idx, t + 1])
    its = out["iterations"].cpu().numpy()
    assert its.min() >= 0 and its.max() <= 99
    assert float((costs[:, -1] < 1.0).float().mean()) > 0.5                      # most instances reach the goal
