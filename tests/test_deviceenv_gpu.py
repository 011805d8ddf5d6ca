"""``tfmpc.envs.deviceenv.DeviceEnv``: an env given as C++ device functions, compiled at run time into the wave-per-instance iLQR kernels with
derivatives by forward-mode dual numbers (csrc/user_env.h) -- SURVEY.md 8(f) N2 at hot-path speed; the reference differentiates whatever
transition / cost it is handed (/root/reference/tfmpc/envs/diffenv.py:13-101).

Validated the way VERDICT round 4 item 6 asks: Navigation and Reservoir WRITTEN AS DeviceEnv source (tests/deviceenv_sources.py) against the
built-in kernels (closed-form derivatives, csrc/envs.h) and the restatement (oracle/envs_ref.py: torch autodiff, fp64) -- all 13 derivative
tensors, rollouts, whole solves."""

import numpy as np
import pytest
import torch

import deviceenv_sources as sources
import problems
from oracle import envs_ref, ilqr_ref
from tfmpc import _hip
from tfmpc.envs import deviceenv
from tfmpc.envs.deviceenv import DeviceEnv
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

needs_hipcc = pytest.mark.skipif(deviceenv.hipcc_path() is None, reason="a DeviceEnv is compiled with hipcc when it is first used")


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


@needs_hipcc
def test_the_companion_library_builds_without_a_gpu_and_exports_the_twins():
    import ctypes
    path = deviceenv.build(sources.NAVIGATION, 2, 2)
    lib = ctypes.CDLL(path)
    for twin in deviceenv._TWINS.values():
        assert hasattr(lib, twin), twin
    assert lib.tfmpc_userenv_state_size() == 2 and lib.tfmpc_userenv_action_size() == 2
    assert deviceenv.build(sources.NAVIGATION, 2, 2) == path                     # cached by source hash
    with pytest.raises(RuntimeError, match="does not compile"):
        deviceenv.build("template <class S> __device__ S cost(const float *p, const S *x, const S *u) { return undeclared; }", 2, 2)


def _nav_pair():
    cfg = problems.NAV_CONFIG
    builtin = Navigation.load(cfg)
    user = DeviceEnv(sources.NAVIGATION, 2, 2, params=sources.navigation_params(cfg), low=np.array(cfg["low"]), high=np.array(cfg["high"]))
    oenv = envs_ref.Navigation(cfg["goal"], cfg["deceleration"]["center"], cfg["deceleration"]["decay"], cfg["low"], cfg["high"])
    return cfg, builtin, user, oenv


def _res_pair(n=4):
    cfg = dict(problems.RES4_CONFIG) if n == 4 else dict(problems.reservoir_config(n, seed=5))
    builtin = Reservoir.load(dict(cfg))
    user = DeviceEnv(sources.reservoir_source(n), n, n, params=sources.reservoir_params(cfg), low=0.0, high=1.0)
    return cfg, builtin, user, envs_ref.Reservoir(**cfg)


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("which", ["navigation", "reservoir4", "reservoir7"])
def test_all_thirteen_derivative_tensors_against_the_builtin_kernels_and_the_autodiff_restatement(which):
    cfg, builtin, user, oenv = _nav_pair() if which == "navigation" else _res_pair(int(which[-1]))
    n, m = user.state_size, user.action_size
    rng = np.random.default_rng(3)
    B, T = 12, 9
    x = (rng.uniform(0, 10, size=(B, T + 1, n, 1)) if which == "navigation" else rng.uniform(20, 95, size=(B, T + 1, n, 1))).astype(np.float32)
    u = (rng.uniform(-1, 1, size=(B, T, m, 1)) if which == "navigation" else rng.uniform(0, 1, size=(B, T, m, 1))).astype(np.float32)
    got = iLQR(user).derivatives(x, u)
    ref = iLQR(builtin).derivatives(x, u)
    torch.cuda.synchronize()
    names = [f"{t}.{f}" for t, tup in zip("tcf", got) for f in tup._fields]
    for name, a, b in zip(names, [v for tup in got for v in tup], [v for tup in ref for v in tup]):
        a, b = _np(a), _np(b)
        assert a.shape == b.shape, name
        scale = max(np.abs(b).max(), 1.0)
        assert np.abs(a - b).max() <= 2e-5 * scale, (which, name, np.abs(a - b).max(), scale)
    # ... and against autodiff in fp64 (the reference's own mechanism), one instance
    x64, u64 = x[0].astype(np.float64), u[0].astype(np.float64)
    tm, cm, fm = oenv.get_linear_transition(x64[:-1], u64), oenv.get_quadratic_cost(x64[:-1], u64), oenv.get_quadratic_final_cost(x64[-1])
    for name, a, b in zip(names, [v for tup in got for v in tup], list(tm) + list(cm) + list(fm)):
        a, b = _np(a)[0], np.asarray(b, dtype=np.float64)
        scale = max(np.abs(b).max(), 1.0)
        assert np.abs(a.reshape(b.shape) - b).max() <= 2e-5 * scale, (which, name)
    # the env protocol itself (diffenv.py): single-step transition / cost / final_cost
    xn_u, xn_b = user.transition(x[:, 0], u[:, 0], batch=True), builtin.transition(x[:, 0], u[:, 0], batch=True)
    assert np.abs(_np(xn_u) - _np(xn_b)).max() <= 2e-5 * max(np.abs(_np(xn_b)).max(), 1.0)
    assert np.abs(_np(user.cost(x[:, 0], u[:, 0], batch=True)) - _np(builtin.cost(x[:, 0], u[:, 0], batch=True))).max() <= 1e-4 * np.abs(_np(builtin.cost(x[:, 0], u[:, 0], batch=True))).max() + 1e-5
    assert np.abs(_np(user.final_cost(x[:, 0], batch=True)) - _np(builtin.final_cost(x[:, 0], batch=True))).max() <= 1e-4 * np.abs(_np(builtin.final_cost(x[:, 0], batch=True))).max() + 1e-5


@pytest.mark.gpu
@needs_hipcc
def test_navigation_whole_solves_match_the_builtin_kernel_and_the_restatement():
    """configs[3] as a DeviceEnv: same iterations and trajectories as the built-in Navigation kernels on (nearly) every instance -- the two
    programs round the zone factor differently (hardware sqrt / exp2 there, libm-accurate here), so a near-tie may flip -- and the fp64
    restatement's final cost inside the usual budget; the decision trace is recorded like a built-in env's."""
    cfg, builtin, user, oenv = _nav_pair()
    rng = np.random.default_rng(4)
    B, T = 256, 50
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = np.stack([problems.scalar_uniform_actions(T, [-1, -1], [1, 1], rng) for _ in range(B)]).astype(np.float32)
    s_user, s_builtin = iLQR(user), iLQR(builtin)
    out = s_user.solve_device(x0, T, u_init=u0, trace_rows=170)
    ref = s_builtin.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    assert s_user.last_kernel.startswith("lane_group") and "user env" in s_user.last_kernel      # n = m = 2: 16 lanes per instance
    # ... and the generic wave kernel on the same user env (what any other shape runs on): the same solve up to the two kernels' rounding
    user._library().force_wave_kernel(True)
    try:
        wave = s_user.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
        assert s_user.last_kernel.startswith("wave") and "user env" in s_user.last_kernel
    finally:
        user._library().force_wave_kernel(False)
    agree = (wave["iterations"] == out["iterations"]).cpu().numpy()
    assert agree.mean() >= 0.9, agree.mean()
    cw = _np(wave["costs"]).sum(1)
    assert np.median(np.abs(cw - _np(out["costs"]).sum(1)) / np.abs(cw)) <= 1e-5
    assert int((out["status"] & ~_hip.ST_NOT_PD).abs().sum()) == 0
    same = (out["iterations"] == ref["iterations"]).cpu().numpy()
    assert same.mean() >= 0.9, same.mean()
    cu, cb = _np(out["costs"]).sum(1), _np(ref["costs"]).sum(1)
    assert np.abs(cu - cb)[same].max() <= 2e-3 * np.abs(cb).max()
    assert np.median(np.abs(cu - cb) / np.abs(cb)) <= 1e-5
    assert int(out["trace_len"].min()) >= 1 and int(out["trace_len"].max()) <= 170
    for b in (0, 17, 101):
        xs, us, cs, it = ilqr_ref.ILQRRef(oenv).solve(x0[b].astype(np.float64), T, u_init=u0[b].astype(np.float64))
        if it == int(out["iterations"][b]):
            assert abs(cs.sum() - cu[b]) <= 2e-3 * abs(cs.sum()), (b, cs.sum(), cu[b])
    # the solve drives the rollout to the goal within the box (sanity of the whole pipeline)
    assert np.all(np.abs(_np(out["actions"])) <= 1.0 + 1e-6)


@pytest.mark.gpu
@needs_hipcc
def test_reservoir_takes_the_bang_bang_branch_like_the_builtin_env():
    """res4 as a DeviceEnv: the cost's second derivatives are exactly zero through the dual numbers too (max / abs are piecewise linear),
    so V_xx stays 0 and the backward pass takes ilqr.py:140-141's bang-bang branch; first iterations against the built-in
    wave kernel (same selector wherever |Q_u| is clear of rounding)."""
    cfg, builtin, user, oenv = _res_pair(4)
    rng = np.random.default_rng(8)
    B, T = 24, 30
    x0 = (np.array(problems.RES4_X0, dtype=np.float32).reshape(1, 4, 1) * rng.uniform(0.8, 1.2, size=(B, 4, 1))).astype(np.float32)
    u0 = iLQR(builtin).random_actions(T, B, seed=2)
    tm, cm, fm = iLQR(user).derivatives(iLQR(user).start(x0, T, u_init=u0)[0], u0)
    for name in ("l_xx", "l_uu", "l_ux", "l_xu"):
        assert float(getattr(cm, name).abs().max()) == 0.0, name
    assert float(fm.l_xx.abs().max()) == 0.0
    with _hip.option("TFMPC_ILQR_KERNEL", "wave"):
        ref = iLQR(builtin, max_iterations=3).solve_device(x0, T, u_init=u0)
    out = iLQR(user, max_iterations=3).solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    cu, cb = _np(out["costs"]).sum(1), _np(ref["costs"]).sum(1)
    assert int(out["status"].abs().sum()) == 0
    assert np.median(np.abs(cu - cb) / np.abs(cb)) <= 1e-3 and np.all(cu <= _np(iLQR(user).start(x0, T, u_init=u0)[2]).sum(1) * (1 + 1e-6))
    assert np.all((_np(out["actions"]) >= -1e-6) & (_np(out["actions"]) <= 1 + 1e-6))


@pytest.mark.gpu
@needs_hipcc
def test_per_instance_parameters_and_unbounded_actions():
    """params [B, P]: one goal per instance (env batch = solve batch); no bounds: the unconstrained Cholesky controller (ilqr.py:357-362)."""
    cfg = problems.NAV_CONFIG
    B, T = 16, 20
    rng = np.random.default_rng(2)
    base = sources.navigation_params(cfg)
    params = np.tile(base, (B, 1))
    params[:, :2] = rng.uniform(2, 9, size=(B, 2))
    user = DeviceEnv(sources.NAVIGATION, 2, 2, params=params)
    assert user.env_batch_size() == B and not user.action_space.is_bounded()
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = 0.1 * rng.normal(size=(B, T, 2, 1)).astype(np.float32)
    out = iLQR(user, max_iterations=30).solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    for b in (0, 7, 15):
        one_cfg = dict(cfg, goal=params[b, :2].reshape(2, 1).tolist())
        o = ilqr_ref.ILQRRef(envs_ref.Navigation(one_cfg["goal"], cfg["deceleration"]["center"], cfg["deceleration"]["decay"], [-np.inf, -np.inf], [np.inf, np.inf]), max_iterations=30)
        xs, us, cs, it = o.solve(x0[b].astype(np.float64), T, u_init=u0[b].astype(np.float64))
        got = _np(out["costs"][b]).sum()
        assert abs(got - cs.sum()) <= 5e-3 * abs(cs.sum()) + 1e-3, (b, got, cs.sum(), it, int(out["iterations"][b]))


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("n,m,T,bound", [(6, 3, 15, None), (8, 4, 20, 0.4), (16, 8, 20, None)])
def test_lq_env_as_device_source_against_the_builtin_lq_env(n, m, T, bound):
    """A DENSE env of another shape than 2 x 2 (the generic wave kernel; from n = 12 on with the backward products on the matrix cores): the LQ
    env of /root/reference/tfmpc/solvers/lqr.py:36-57 written as DeviceEnv source with per-instance parameters against the built-in LQ env on the
    same wave kernel -- derivatives (here the dual numbers must reproduce F and the symmetric part of C exactly up to rounding) and whole solves,
    unbounded (Cholesky controller) and control-limited (box-QP)."""
    from tfmpc.envs.lq import LQEnv
    B = 24
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=5 * n + m)
    F = F * 0.25 * np.sqrt(16.0 / n)
    low, high = (None, None) if bound is None else (-bound, bound)
    builtin = LQEnv(F, f, C, c, low=low, high=high)
    user = DeviceEnv(sources.lq_source(n, m), n, m, params=sources.lq_params(F, f, C, c), low=low, high=high)
    rng = np.random.default_rng(1)
    u0 = np.clip(0.1 * rng.normal(size=(B, T, m, 1)), -(bound or 1.0), bound or 1.0).astype(np.float32)
    x0 = x0.astype(np.float32)[..., None]
    xs = iLQR(builtin).start(x0, T, u_init=u0)[0]
    got, ref = iLQR(user).derivatives(xs, u0), iLQR(builtin).derivatives(xs, u0)
    torch.cuda.synchronize()
    for a, b_ in zip([v for tup in got for v in tup], [v for tup in ref for v in tup]):
        a, b_ = _np(a), _np(b_)
        assert np.abs(a - b_).max() <= 2e-5 * max(np.abs(b_).max(), 1.0)
    with _hip.option("TFMPC_ILQR_KERNEL", "wave"):
        ref_out = iLQR(builtin, max_iterations=8).solve_device(x0, T, u_init=u0)
    out = iLQR(user, max_iterations=8).solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    assert int((out["status"] & ~_hip.ST_NOT_PD).abs().sum()) == 0
    same = (out["iterations"] == ref_out["iterations"]).cpu().numpy()
    assert same.mean() >= 0.8, same.mean()
    cu, cb = _np(out["costs"]).sum(1), _np(ref_out["costs"]).sum(1)
    assert np.median(np.abs(cu - cb) / np.abs(cb)) <= 1e-4, np.median(np.abs(cu - cb) / np.abs(cb))
    assert np.abs(_np(out["states"]) - _np(ref_out["states"]))[same].max() <= 5e-3 * np.abs(_np(ref_out["states"])).max()


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("n,m,T,bound,B", [(2, 2, 6, None, 4), (2, 2, 30, 0.5, 300), (2, 2, 25, None, 2500), (2, 1, 50, 0.3, 64), (1, 1, 12, None, 9),
                                           (3, 1, 30, 0.5, 300), (1, 2, 20, 0.4, 2500), (1, 3, 16, None, 100), (2, 1, 40, None, 2100)])
def test_tiny_user_envs_take_a_lane_group_kernel_without_spills(n, m, T, bound, B):
    """Round 6: the LQ env of lqr.py:36-57 as a 2 x 2 DeviceEnv source needs ~200 vector registers in the lane-group kernel; at that kernel's usual
    budget (three waves per SIMD: 168) it spilled 37 of them, and the spilled instantiation lost scalars -- the first case below HUNG, larger batches
    returned garbage (csrc/ilqr_lane_kernels.h).  The launcher now asks the runtime for an instantiation without a private segment (one wave's budget
    here).  The same round opened the kernel to every dense env with n + m <= 4 (a pendulum is 2 x 1): the other shapes below.  Against the same env
    on the generic wave kernel and the built-in LQ env."""
    from tfmpc.envs.lq import LQEnv
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=16)
    F = F * 0.3
    low, high = (None, None) if bound is None else (-bound, bound)
    builtin = LQEnv(F, f, C, c, low=low, high=high)
    user = DeviceEnv(sources.lq_source(n, m), n, m, params=sources.lq_params(F, f, C, c), low=low, high=high)
    rng = np.random.default_rng(3)
    u0 = np.clip(0.1 * rng.normal(size=(B, T, m, 1)), -(bound or 1.0), bound or 1.0).astype(np.float32)
    x0 = x0.astype(np.float32)[..., None]
    s_user = iLQR(user, max_iterations=8)
    out = s_user.solve_device(x0, T, u_init=u0, trace_rows=30)
    torch.cuda.synchronize()
    assert s_user.last_kernel.startswith("lane_group") and "user env" in s_user.last_kernel
    out = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items() if k != "workspace"}
    user._library().force_wave_kernel(True)
    try:
        wave = s_user.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
        assert s_user.last_kernel.startswith("wave")
    finally:
        user._library().force_wave_kernel(False)
    with _hip.option("TFMPC_ILQR_KERNEL", "wave"):
        ref = iLQR(builtin, max_iterations=8).solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    assert int((out["status"] & ~_hip.ST_NOT_PD).abs().sum()) == 0
    cu = _np(out["costs"]).sum(1)
    for other in (wave, ref):
        same = (out["iterations"] == other["iterations"]).cpu().numpy()
        assert same.mean() >= 0.8, same.mean()
        co = _np(other["costs"]).sum(1)
        assert np.median(np.abs(cu - co) / np.abs(co)) <= 1e-4, np.median(np.abs(cu - co) / np.abs(co))
        assert np.abs(_np(out["states"]) - _np(other["states"]))[same].max() <= 5e-3 * np.abs(_np(other["states"])).max()
    if bound is not None:
        assert np.all(np.abs(_np(out["actions"])) <= bound + 1e-6)
    assert int(out["trace_len"].min()) >= 1


PENDULUM = sources.PENDULUM



@pytest.mark.gpu
@needs_hipcc
def test_a_pendulum_written_as_device_source_runs_the_lane_group_kernel():
    """The canonical "env of your own": a torque-limited pendulum (n = 2, m = 1, smooth dense cost) as DeviceEnv source -- lane-group kernel (round
    6: n + m <= 4) against the generic wave kernel on the same source, and the solve does what iLQR is for: the cost falls well below the start's
    on every instance, inside the torque limits, with no flag raised."""
    B, T = 512, 60
    rng = np.random.default_rng(5)
    params = np.array([0.05, 9.81, 0.1, 0.0, 1.0, 0.1, 0.01], dtype=np.float32)
    user = DeviceEnv(PENDULUM, 2, 1, params=params, low=-4.0, high=4.0)
    x0 = np.stack([rng.uniform(-1.2, 1.2, size=B), rng.uniform(-1.0, 1.0, size=B)], axis=1).astype(np.float32)[..., None]
    u0 = np.zeros((B, T, 1, 1), dtype=np.float32)
    s = iLQR(user, max_iterations=30)
    start_cost = _np(s.start(x0, T, u_init=u0)[2]).sum(1)
    out = s.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    assert s.last_kernel.startswith("lane_group")
    out = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items() if k != "workspace"}
    user._library().force_wave_kernel(True)
    try:
        wave = s.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
        assert s.last_kernel.startswith("wave")
    finally:
        user._library().force_wave_kernel(False)
    cu, cw = _np(out["costs"]).sum(1), _np(wave["costs"]).sum(1)
    assert int((out["status"] & ~_hip.ST_NOT_PD).abs().sum()) == 0
    assert np.median(np.abs(cu - cw) / np.abs(cw)) <= 1e-4 and (out["iterations"] == wave["iterations"]).float().mean() >= 0.8
    assert np.all(cu <= start_cost * (1 + 1e-6)) and np.median(cu / start_cost) < 0.5
    assert np.all(np.abs(_np(out["actions"])) <= 4.0 + 1e-6)
