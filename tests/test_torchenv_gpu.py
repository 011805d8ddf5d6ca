"""Generic differentiable envs given as torch functions (the reference's
"any TensorFlow-differentiable env", tfmpc/envs/diffenv.py:13-101): torch.func derivatives +
the HIP Riccati backward pass + batched torch rollouts must reproduce what the built-in
device-resident envs produce for the same model."""

import numpy as np
import pytest
import torch

import problems
from tfmpc.envs.lqr.navigation import NavigationLQR
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.torchenv import TorchEnv
from tfmpc.solvers.ilqr import iLQR

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _host_driven_loop():
    """This file covers the HOST-driven loop of a generic env (torch.func derivatives, HIP backward pass, torch rollouts): since round 6 iLQR would
    translate these functions to device source instead (tests/test_fxenv_gpu.py covers that), so the automatic translation is switched off here."""
    before = TorchEnv.auto_compile
    TorchEnv.auto_compile = False
    yield
    TorchEnv.auto_compile = before


def _navlqr_torch(goal, beta, low=None, high=None):
    g = torch.as_tensor(goal, dtype=torch.float32, device="cuda").reshape(-1)
    return TorchEnv(lambda x, u: x + u,
                    lambda x, u: ((x - g) ** 2).sum() + beta * (u ** 2).sum(),
                    lambda x: ((x - g) ** 2).sum(), 2, 2, low, high)


def _navigation_torch(cfg):
    g = torch.as_tensor(cfg["goal"], dtype=torch.float32, device="cuda").reshape(-1)
    centers = torch.as_tensor(cfg["deceleration"]["center"], dtype=torch.float32, device="cuda").reshape(-1, 2)
    decay = torch.as_tensor(cfg["deceleration"]["decay"], dtype=torch.float32, device="cuda")

    def transition(x, u):
        dist = torch.linalg.norm(x[None, :] - centers, dim=-1)
        lam = torch.prod(2.0 / (1.0 + torch.exp(-decay * dist)) - 1.0)
        return x + lam * u

    cost = lambda x, u: ((x - g) ** 2).sum()
    return TorchEnv(transition, cost, lambda x: ((x - g) ** 2).sum(), 2, 2,
                    np.asarray(cfg["low"]).reshape(2, 1), np.asarray(cfg["high"]).reshape(2, 1))


def test_torch_func_models_match_the_device_closed_forms():
    cfg = problems.NAV_CONFIG
    builtin, generic = Navigation.load(cfg), _navigation_torch(cfg)
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 10, size=(9, 2, 1)).astype(np.float32)
    u = rng.uniform(-1, 1, size=(9, 2, 1)).astype(np.float32)
    a, b = builtin.get_linear_transition(x, u), generic.get_linear_transition(x, u)
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
    a, b = builtin.get_quadratic_cost(x, u), generic.get_quadratic_cost(x, u)
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
    a, b = builtin.get_quadratic_final_cost(x[0]), generic.get_quadratic_final_cost(x[0])
    for p, q in zip(a, b):
        assert torch.allclose(p, q.reshape(p.shape), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bounds", [None, (-1.0, 1.0)])
def test_generic_env_solve_equals_builtin_on_linear_navigation(bounds):
    low, high = bounds if bounds else (None, None)
    goal, beta, B, T = [[5.5], [-9.0]], 5.0, 12, 10
    sb, sg = iLQR(NavigationLQR(goal, beta, low, high)), iLQR(_navlqr_torch(goal, beta, low, high))
    x0 = np.random.default_rng(2).normal(size=(B, 2, 1)).astype(np.float32)
    u0 = sb.random_actions(T, B, seed=4)
    tb, ib = sb.solve(x0, T, u_init=u0)
    tg, ig = sg.solve(x0, T, u_init=u0)
    assert np.array_equal(ib, ig)
    assert np.abs(tb.states - tg.states).max() <= 1e-3 * np.abs(tb.states).max()
    assert np.abs(tb.costs - tg.costs).max() <= 1e-3 * np.abs(tb.costs).max()
    # the piecewise API works on generic envs too (reference tests/test_ilqr.py:48-109)
    xs, us, cs = sg.start(x0[0], T, u_init=u0[0])
    models = sg.derivatives(xs, us)
    K, k, J, dV1, dV2 = sg.backward(T, us, *models)
    st, ac, co, Jn, res = sg.forward(xs, us, K, k)
    assert st.shape == xs.shape and ac.shape == us.shape and co.shape == (T + 1,)
    assert torch.allclose(st[1], sg.env.transition(st[0], ac[0]))


def test_generic_env_solve_tracks_builtin_on_nonlinear_navigation():
    cfg = problems.NAV_CONFIG
    sb, sg = iLQR(Navigation.load(cfg)), iLQR(_navigation_torch(cfg))
    B, T = 24, 20
    x0 = np.random.default_rng(3).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = sb.random_actions(T, B, seed=6)
    tb, ib = sb.solve(x0, T, u_init=u0)
    tg, ig = sg.solve(x0, T, u_init=u0)
    rel = np.abs(tb.total_cost - tg.total_cost) / np.abs(tb.total_cost)
    assert np.median(rel) <= 1e-4 and np.quantile(rel, 0.9) <= 2e-2       # a few flipped line searches
    assert np.mean(ib == ig) >= 0.7
    assert np.abs(tg.actions).max() <= 1.0


# ---- HVAC and Reservoir written as plain torch functions, against the ORACLE (oracle/envs_ref.py = the reference's
# ---- env equations differentiated by autodiff, oracle/ilqr_ref.py = ilqr.py), not against the built-in kernels --------
def _hvac_torch(cfg):
    """tfmpc/envs/hvac/__init__.py:69-149 for one instance: x[n], u[n]."""
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device="cuda")
    vec = lambda key: t(cfg[key]).reshape(-1)
    adj = np.asarray(cfg["adj"], dtype=bool)
    G = t(np.logical_or(adj, adj.T).astype(np.float32)) / t(cfg["R_wall"])
    k_out = t(np.asarray(cfg["adj_outside"], dtype=np.float32)).reshape(-1) / vec("R_outside")
    k_hall = t(np.asarray(cfg["adj_hall"], dtype=np.float32)).reshape(-1) / vec("R_hall")
    t_out, t_hall, cap, air_max = vec("temp_outside"), vec("temp_hall"), vec("capacity"), vec("air_max")
    lo, hi = vec("temp_lower_bound"), vec("temp_upper_bound")
    n = lo.numel()

    def transition(x, u):
        heating = u * air_max * 1.006 * (40.0 - x)
        between = -(G * (x[:, None] - x[None, :])).sum(dim=-1)
        return x + 1.0 / cap * (heating + between + k_out * (t_out - x) + k_hall * (t_hall - x))

    def penalties(x):
        return (20000.0 * (torch.relu(lo - x) + torch.relu(x - hi)) + 10.0 * torch.abs((lo + hi) / 2 - x)).sum()

    return TorchEnv(transition, lambda x, u: (u * air_max).sum() + penalties(x), penalties, n, n, 0.0, 1.0)


def _reservoir_torch(cfg):
    """tfmpc/envs/reservoir/__init__.py:47-105 (cec=True) for one instance."""
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device="cuda")
    vec = lambda key: t(cfg[key]).reshape(-1)
    cap, lo, hi = vec("max_res_cap"), vec("lower_bound"), vec("upper_bound")
    lp, hp, sp = vec("low_penalty"), vec("high_penalty"), vec("set_point_penalty")
    rain, D = vec("rain_shape") * vec("rain_scale"), t(cfg["downstream"])
    n = lo.numel()

    def transition(x, u):
        out = u * x
        return x + rain + D.T @ out - 0.5 * torch.sin(x / cap) * x - out

    def cost(x, u=None):
        return (-lp * torch.relu(lo - x) - hp * torch.relu(x - hi) - sp * torch.abs((lo + hi) / 2.0 - x)).sum()

    return TorchEnv(transition, cost, cost, n, n, 0.0, 1.0)


def _oracle_pair(kind):
    from oracle import envs_ref
    if kind == "hvac":
        cfg = dict(problems.HVAC6_CONFIG)
        return cfg, _hvac_torch(cfg), envs_ref.HVAC(**cfg), np.asarray(problems.HVAC6_X0, dtype=np.float32)
    cfg = dict(problems.RES4_CONFIG)
    return cfg, _reservoir_torch(cfg), envs_ref.Reservoir(**cfg), np.asarray(problems.RES4_X0, dtype=np.float32)


@pytest.mark.parametrize("kind", ["hvac", "reservoir"])
def test_generic_env_pieces_match_the_oracle(kind):
    """start / derivatives / backward / forward of the reference's API (tests/test_ilqr.py:48-109) on an env given as
    torch functions, each piece against the fp64 oracle on the oracle's own inputs (so a flipped bang-bang decision
    cannot leak from one piece into the next)."""
    from oracle import ilqr_ref
    cfg, genv, oenv, x0 = _oracle_pair(kind)
    n, T = genv.state_size, 12
    sg, o = iLQR(genv), ilqr_ref.ILQRRef(oenv)
    u0 = problems.scalar_uniform_actions(T, o.low, o.high, np.random.default_rng(9)).astype(np.float32)
    xs, us, cs = o.start(x0, T, u_init=u0)
    gx, gu, gc = sg.start(x0, T, u_init=u0)
    assert np.abs(gx.cpu().numpy() - xs).max() <= 2e-5 * np.abs(xs).max()
    assert np.abs(gc.cpu().numpy() - cs).max() <= 2e-5 * np.abs(cs).max()
    # derivatives: torch.func on the user's functions vs autodiff of the oracle's env restatement (diffenv.py:13-101)
    tm, cm, fm = o.derivatives(xs, us)
    gtm, gcm, gfm = sg.derivatives(xs.astype(np.float32), us.astype(np.float32))
    for got, ref, name in [(gtm.f, tm.f, "f"), (gtm.f_x, tm.f_x, "f_x"), (gtm.f_u, tm.f_u, "f_u"), (gcm.l, cm.l, "l"),
                           (gcm.l_x, cm.l_x, "l_x"), (gcm.l_u, cm.l_u, "l_u"), (gcm.l_xx, cm.l_xx, "l_xx"),
                           (gcm.l_uu, cm.l_uu, "l_uu"), (gcm.l_ux, cm.l_ux, "l_ux"), (gcm.l_xu, cm.l_xu, "l_xu"),
                           (gfm.l, fm.l, "fl"), (gfm.l_x, fm.l_x, "fl_x"), (gfm.l_xx, fm.l_xx, "fl_xx")]:
        g = got.cpu().numpy().reshape(np.shape(ref))
        assert np.abs(g - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1.0), name
    # backward on the oracle's models (HIP Riccati kernel, bang-bang branch: V_xx == 0 on these envs)
    K, k, J, dV1, dV2 = o.backward(T, us, tm, cm, fm, mu=0.0)
    gK, gk, gJ, g1, g2 = sg.backward(T, us.astype(np.float32), gtm, gcm, gfm, mu=0.0)
    assert not K.any() and not bool(gK.any())
    # same bound for every action whose Q_u is not a near-tie (Q_u = l_u + f_u^T V_x with V_x <- Q_x, since K == 0:
    # Reservoir's Q_u,i = x_i (V_x,i+1 - V_x,i) cancels to rounding wherever two costates coincide)
    V_x, clear, Q_u_all = fm.l_x, np.zeros(k.shape, dtype=bool), np.zeros(k.shape)
    for t in range(T - 1, -1, -1):
        Q_u = Q_u_all[t] = cm.l_u[t] + tm.f_u[t].T @ V_x
        clear[t] = np.abs(Q_u) > 1e-3 * (np.abs(cm.l_u[t]) + np.abs(tm.f_u[t]).T @ np.abs(V_x) + 1e-300)
        V_x = cm.l_x[t] + tm.f_x[t].T @ V_x
    assert clear.mean() > 0.8
    assert np.array_equal(np.sign(gk.cpu().numpy())[clear], np.sign(k)[clear])
    k = gk.cpu().numpy().astype(np.float64)         # the pieces below are driven with the device's own selector
    dV1 = float((k * Q_u_all).sum())                  # ilqr.py:166 for that selector
    assert abs(float(g1) - dV1) <= 1e-4 * abs(dV1)
    assert abs(float(gJ) - J) <= 2e-5 * abs(J) and float(g2) == dV2 == 0.0
    # forward with the oracle's gains at two step sizes (ilqr.py:174-212)
    for alpha in (1.0, 0.0631):
        x1, u1, c1, J1, r1 = o.forward(xs, us, K, k, alpha)
        fx, fu, fc, fJ, fr = sg.forward(xs.astype(np.float32), us.astype(np.float32), K.astype(np.float32), k.astype(np.float32), alpha)
        assert np.abs(fx.cpu().numpy() - x1).max() <= 5e-5 * np.abs(x1).max(), alpha
        assert np.abs(fu.cpu().numpy() - u1).max() <= 1e-6 and abs(float(fr) - r1) <= 1e-6
        assert abs(float(fJ) - J1) <= 5e-5 * abs(J1)


def test_generic_hvac_two_iterations_match_the_oracle():
    """Two whole iterations of the host-driven loop (ilqr.py:214-283) on HVAC-as-torch-functions against the fp64
    restatement; the fp32 restatement sets the budget (its decisions are clear-cut on this env)."""
    from oracle import envs_ref, ilqr_ref
    cfg, genv, oenv, _ = _oracle_pair("hvac")
    T, B = 12, 5
    rng = np.random.default_rng(4)
    x0 = rng.uniform(5.0, 30.0, size=(B, 6, 1)).astype(np.float32)
    sg = iLQR(genv, max_iterations=2)
    u0 = sg.random_actions(T, B, seed=2)
    tg, ig = sg.solve(x0, T, u_init=u0)
    o64 = ilqr_ref.ILQRRef(oenv, max_iterations=2)
    o32 = ilqr_ref.ILQRRef(envs_ref.HVAC(**cfg, dtype=np.float32), dtype=np.float32, max_iterations=2)
    for b in range(B):
        ub = u0[b].cpu().numpy()
        x64, u64, c64, it64 = o64.solve(x0[b], T, u_init=ub)
        x32, u32, c32, it32 = o32.solve(x0[b], T, u_init=ub)
        assert int(ig[b]) == it64
        for got, r64, r32, what in ((tg.states[b], x64, x32, "states"), (tg.actions[b], u64, u32, "actions"), (tg.costs[b], c64, c32, "costs")):
            allowed = 5 * max(np.abs(r32.astype(np.float64) - r64).max(), 1e-6 * np.abs(r64).max())
            assert np.abs(got - r64).max() <= allowed, (b, what, np.abs(got - r64).max(), allowed)


def test_graph_replay_of_the_generic_path_changes_nothing():
    """`iLQR(env, graphs=True)` (the default) replays the rollout and derivative blocks of the host-driven loop as hipGraphs:
    same launches on the same inputs, so every output must equal the eager loop's bit for bit -- also when the batch
    changes between solves (a new capture per shape)."""
    cfg = problems.NAV_CONFIG
    env = _navigation_torch(cfg)
    rng = np.random.default_rng(3)
    for B, T in ((33, 12), (7, 9), (33, 12)):
        x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
        outs = {}
        for graphs in (False, True):
            solver = outs.setdefault(("solver", graphs), iLQR(env, max_iterations=6, graphs=graphs))
            u0 = solver.random_actions(T, B, seed=B)
            outs[graphs] = solver.solve_device(x0, T, u_init=u0)
        assert not any(g.eager for g in outs[("solver", True)]._graphed.values())      # both blocks were captured
        for key in ("states", "actions", "costs", "iterations", "status"):
            assert torch.equal(outs[False][key], outs[True][key]), (B, T, key)


def test_an_env_that_cannot_be_captured_runs_eagerly():
    """A transition function that copies from the host on every call cannot be recorded into a graph: the block falls back to
    eager execution (once, remembered) and the solve returns what the eager solver returns."""
    g_host = np.array([8.0, 9.0], dtype=np.float32)

    def transition(x, u):
        return x + u + 0.0 * torch.as_tensor(g_host, device=x.device).sum()          # host-to-device copy inside the env

    g = torch.as_tensor(g_host, device="cuda")
    env = TorchEnv(transition, lambda x, u: ((x - g) ** 2).sum() + 0.1 * (u ** 2).sum(), lambda x: ((x - g) ** 2).sum(), 2, 2)
    x0 = np.random.default_rng(1).uniform(0, 10, size=(5, 2, 1)).astype(np.float32)
    outs = {}
    for graphs in (False, True):
        solver = iLQR(env, max_iterations=3, graphs=graphs)
        outs[graphs] = solver.solve_device(x0, 6, u_init=solver.random_actions(6, 5, seed=2))
        if graphs:
            assert solver._graphed["rollouts"].eager
    for key in ("states", "actions", "costs", "iterations"):
        assert torch.equal(outs[False][key], outs[True][key]), key


def test_rebinding_what_the_env_functions_close_over_drops_the_captured_graphs():
    """a captured hipGraph has baked in the ADDRESS of every tensor (and the value of every Python scalar) the env's
    functions close over.  Rebinding one of them between two solves of the same shape must not replay the stale graph: `solve`
    compares a signature of the closures (tensor identity + address, scalars by value, the bounds by value) and captures again."""
    def make():
        goal = torch.tensor([8.0, 9.0], device="cuda")
        weight = 0.1

        def cost(x, u):
            return ((x - goal) ** 2).sum() + weight * (u ** 2).sum()

        def final(x):
            return ((x - goal) ** 2).sum()

        def rebind(new_goal, new_weight):
            nonlocal goal, weight
            goal, weight = new_goal, new_weight
        return TorchEnv(lambda x, u: x + u, cost, final, 2, 2), rebind

    env, rebind = make()
    B, T = 9, 8
    x0 = np.random.default_rng(5).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=4)                   # graphs=True is the default
    u0 = solver.random_actions(T, B, seed=1)
    first = {k: v.clone() for k, v in solver.solve_device(x0, T, u_init=u0).items() if torch.is_tensor(v)}
    assert not any(g.eager for g in solver._graphed.values()) and len(solver._graphed) == 2
    captured = dict(solver._graphed)
    again = solver.solve_device(x0, T, u_init=u0)          # nothing changed: the same graphs replay
    assert all(solver._graphed[k] is captured[k] for k in captured) and torch.equal(again["states"], first["states"])
    rebind(torch.tensor([-3.0, 2.0], device="cuda"), 0.5)  # a NEW tensor and another scalar behind the same function objects
    second = solver.solve_device(x0, T, u_init=u0)
    assert all(solver._graphed[k] is not captured[k] for k in captured)          # captured again
    eager = iLQR(env, max_iterations=4, graphs=False).solve_device(x0, T, u_init=u0)
    for key in ("states", "actions", "costs", "iterations"):
        assert torch.equal(second[key], eager[key]), key
    assert not torch.equal(second["states"], first["states"])
    # an in-place update keeps the address: the replay sees it, no new capture is needed
    captured = dict(solver._graphed)
    env._l.__closure__                                          # (the cells are what the signature reads)
    for cell in env._l.__closure__:
        if isinstance(cell.cell_contents, torch.Tensor):
            cell.cell_contents.copy_(torch.tensor([1.0, 1.0], device="cuda"))
    third = solver.solve_device(x0, T, u_init=u0)
    assert all(solver._graphed[k] is captured[k] for k in captured)
    eager = iLQR(env, max_iterations=4, graphs=False).solve_device(x0, T, u_init=u0)
    assert torch.equal(third["states"], eager["states"])
    # new bounds on the same env object: the cached device copies go with the graphs
    env.action_space.low[:] = -0.25
    env.action_space.high[:] = 0.25
    fourth = solver.solve_device(x0, T, u_init=u0)
    assert float(fourth["actions"].abs().max()) <= 0.25 + 1e-6


def test_decision_trace_of_the_generic_env_path():
    """`iLQR.solve(trace=True)` on a TorchEnv (it raised KeyError): the host-driven loop records the same row per pass
    as the fused kernels -- compared with the built-in NavigationLQR's device trace on the same problems (same decisions, numbers
    to fp32 accuracy)."""
    from tfmpc.solvers.ilqr import trace_records
    goal, beta, B, T = [[5.5], [-9.0]], 5.0, 10, 10
    sb, sg = iLQR(NavigationLQR(goal, beta, -1.0, 1.0)), iLQR(_navlqr_torch(goal, beta, -1.0, 1.0))
    x0 = np.random.default_rng(2).normal(size=(B, 2, 1)).astype(np.float32)
    u0 = sb.random_actions(T, B, seed=4)
    ob = sb.solve_device(x0, T, u_init=u0, trace_rows=40)
    og = sg.solve_device(x0, T, u_init=u0, trace_rows=40)
    assert torch.equal(ob["trace_len"], og["trace_len"]) and torch.equal(ob["iterations"], og["iterations"])
    for rb, rg in zip(trace_records(ob["trace"], ob["trace_len"]), trace_records(og["trace"], og["trace_len"])):
        assert len(rb) == len(rg) >= 1
        for a, c in zip(rb, rg):
            assert (a["iteration"], a["alpha_index"], a["accepted"]) == (c["iteration"], c["alpha_index"], c["accepted"])
            for key in ("mu", "delta", "J_hat", "g_norm", "J", "residual"):
                if a[key] is not None:
                    assert abs(a[key] - c[key]) <= 1e-3 * max(abs(a[key]), 1e-3), (key, a[key], c[key])
    traj, it = sg.solve(x0[0], T, show_progress=False, u_init=u0[0], trace=True)          # the call that used to raise
    assert len(sg.last_trace[0]) >= it + 1
