"""Python env -> DeviceEnv source, automatically (round 6): ``TorchEnv.to_device_env()`` traces the user's torch
functions and emits the three device templates (tfmpc/envs/fxsource.py).  The reference differentiates whatever Python transition / cost it is
handed (/root/reference/tfmpc/envs/diffenv.py:13-101); here Navigation, Reservoir and HVAC WRITTEN AS PLAIN TORCH FUNCTIONS (tests/torch_envs.py)
are translated and held against the built-in kernels (closed forms, csrc/envs.h), torch.func autodiff of the very same functions, and the fp64
restatement -- all 13 derivative tensors, rollouts, whole solves."""

import numpy as np
import pytest
import torch

import problems
import torch_envs
from tfmpc import _hip
from tfmpc.envs import deviceenv, fxsource
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.navigation import Navigation
from tfmpc.envs.reservoir import Reservoir
from tfmpc.envs.torchenv import TorchEnv
from tfmpc.solvers.ilqr import iLQR

needs_hipcc = pytest.mark.skipif(deviceenv.hipcc_path() is None, reason="a DeviceEnv is compiled with hipcc when it is first used")


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


# ---- the translator alone (no GPU) -------------------------------------------------------------------------------------------------------------

def test_translation_of_the_three_envs_and_where_their_constants_go():
    nav = torch_envs.navigation(problems.NAV_CONFIG).to_device_env()
    assert nav.state_size == 2 and nav.action_size == 2 and nav.params.shape == (8,)           # 2 centres x 2, 2 decays, goal (stored once)
    assert "    const float v" not in nav.source and "(-p[" in nav.source                            # (the negated decays: a sign on the operand, no statement)
    assert "sqrt(" in nav.source and "exp(" in nav.source and "x_next[1]" in nav.source
    res = torch_envs.reservoir(problems.RES4_CONFIG).to_device_env()
    assert "sin(" in res.source and "abs(" in res.source and "max(0.0f" in res.source           # relu -> max(0, y): the tie goes to the constant
    assert res.action_space.is_bounded()
    hv = torch_envs.hvac(problems.hvac_config(6)).to_device_env()
    # arithmetic BETWEEN constants (adj / R_wall, dt / capacity, (lo + hi) / 2 ...) is done at translation time and lands in derived parameter slots
    assert "    const float v" not in hv.source and hv.params.shape[0] > 6 * 11 + 36
    # the translator PROVES piecewise-affine costs (sums, products with constants, relu / abs of such): those envs get the costate form of the kernels
    assert res.zero_cost_hessian and hv.zero_cost_hessian and not nav.zero_cost_hessian
    unbounded = torch_envs.reservoir(problems.RES4_CONFIG)
    unbounded.action_space = type(unbounded.action_space)(-np.inf, np.inf, (4, 1))
    assert not unbounded.to_device_env().zero_cost_hessian                      # (unbounded actions: ilqr.py:143 takes the Cholesky controller)
    # a zero of the adjacency mask is structure: the pairs of rooms that do not touch cost no statement
    dense = torch_envs.hvac(dict(problems.hvac_config(6), adj=(~np.eye(6, dtype=bool)).tolist())).to_device_env()
    assert hv.source.count("\n") < dense.source.count("\n")


def test_unsupported_operations_and_python_branches_are_refused_with_a_clear_error():
    good = lambda x: (x ** 2).sum()
    with pytest.raises(fxsource.UnsupportedOperation, match="lgamma"):
        TorchEnv(lambda x, u: torch.lgamma(x) + u, lambda x, u: good(x), good, 2, 2).to_device_env()
    def branchy(x, u):
        if x[0] > 0:                                # a Python branch on the state: a trace would keep one side only
            return x + u
        return x - u
    with pytest.raises(fxsource.UnsupportedOperation, match="torch.where"):
        TorchEnv(branchy, lambda x, u: good(x), good, 2, 2).to_device_env()
    with pytest.raises(ValueError, match="expected 2 values"):
        TorchEnv(lambda x, u: torch.cat([x, u]), lambda x, u: good(x), good, 2, 2).to_device_env()
    with pytest.raises(ValueError, match="expected a scalar"):
        TorchEnv(lambda x, u: x + u, lambda x, u: x * u, good, 2, 2).to_device_env()


def test_in_place_writes_through_views_and_the_other_vocabulary():
    """`out[i] = ...` on a fresh tensor, slices, cat / stack, where / clamp / maximum, matmul with a constant, a detach: everything lands in the
    straight-line program (checked by evaluating the SAME function in torch and a tiny interpreter of the emitted statements)."""
    W = torch.tensor([[0.5, -1.0, 0.0], [2.0, 0.0, 1.5], [0.0, 0.25, -0.75]])

    def transition(x, u):
        out = torch.zeros(3)
        out[0] = x[0] + torch.tanh(u[0])
        out[1:] = (W @ x)[1:] * torch.sigmoid(u[1])
        out = torch.where(out > 1.0, 1.0 + 0.1 * (out - 1.0), out)
        return torch.clamp(out, -5.0, 5.0) + torch.stack([x[2], x[0], x[1]]) * 0.1

    def cost(x, u):
        z = torch.cat([x, u])
        return torch.maximum(z, torch.zeros(5)).sum() + (z ** 3).mean() + torch.sqrt(1.0 + (x * x.detach()).sum())

    src, params = fxsource.translate(transition, cost, lambda x: torch.abs(x).sum(), 3, 2)
    rng = np.random.default_rng(0)
    for _ in range(5):
        x, u = rng.normal(size=3).astype(np.float32), rng.normal(size=2).astype(np.float32)
        got = _evaluate(src, "transition", params, x, u)
        want = transition(torch.as_tensor(x), torch.as_tensor(u)).numpy()
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6), (got, want)
        assert np.isclose(_evaluate(src, "cost", params, x, u), float(cost(torch.as_tensor(x), torch.as_tensor(u))), rtol=1e-5)
        assert np.isclose(_evaluate(src, "final_cost", params, x, None), float(np.abs(x).sum()), rtol=1e-6)


def test_activation_and_statistics_vocabulary():
    """softplus / elu / leaky_relu / silu / logsigmoid / log1p / expm1 / lerp / addcmul / var / std / dist: what a learned-dynamics env is made of."""
    import torch.nn.functional as Fn
    W = torch.tensor([[0.3, -0.2, 0.1], [0.0, 0.5, -0.4], [0.2, 0.0, 0.7]])

    def transition(x, u):
        h = Fn.softplus(W @ x) + Fn.elu(x) * 0.1 + Fn.leaky_relu(x, 0.2) + torch.log1p(x.abs()) - torch.expm1(-x.abs()) * 0.05
        return torch.lerp(x, h, 0.5) + Fn.silu(torch.cat([u, u[:1]]))

    def cost(x, u):
        return torch.var(x) + torch.std(torch.cat([x, u])) + torch.addcmul(x, x, x, value=0.5).sum() + Fn.logsigmoid(u).sum()

    final = lambda x: torch.dist(x, torch.ones(3))
    src, params, info = fxsource.translate_ex(transition, cost, final, 3, 2)
    assert not info["cost_is_piecewise_linear"]
    rng = np.random.default_rng(1)
    for _ in range(4):
        x, u = rng.normal(size=3).astype(np.float32), rng.normal(size=2).astype(np.float32)
        assert np.allclose(_evaluate(src, "transition", params, x, u), transition(torch.as_tensor(x), torch.as_tensor(u)).numpy(), rtol=2e-5, atol=1e-6)
        assert np.isclose(_evaluate(src, "cost", params, x, u), float(cost(torch.as_tensor(x), torch.as_tensor(u))), rtol=2e-5)
        assert np.isclose(_evaluate(src, "final_cost", params, x, None), float(final(torch.as_tensor(x))), rtol=2e-5)
    # piecewise-affine analysis: sums, constants, relu / abs / maximum / where keep it; a product of two inputs or a smooth nonlinearity ends it
    lin = lambda c: fxsource.translate_ex(lambda x, u: x + u, c, lambda x: x.abs().sum(), 2, 2)[2]["cost_is_piecewise_linear"]
    assert lin(lambda x, u: (3.0 * torch.relu(x - 1.0) + torch.abs(u) / 2.0).sum() + torch.maximum(x, u).sum() + torch.where(x > 0, x, -2.0 * x).sum())
    assert not lin(lambda x, u: (x * u).sum()) and not lin(lambda x, u: torch.sqrt(1.0 + x.abs()).sum()) and not lin(lambda x, u: (x ** 2).sum())


def test_vehicle_and_arm_vocabulary():
    """Round 6: tan / atan / asin / acos / atan2 / sinh / cosh / erf -- values of the translated statements against torch (their derivatives: the
    dual-number forms in csrc/user_env.h, held against torch.func on the device in test_the_bicycle_from_python...)."""
    def transition(x, u):
        return torch.stack([torch.tan(0.3 * x[0]) + torch.atan(x[1]), torch.asin(0.5 * torch.tanh(x[2])) + torch.acos(0.4 * torch.sin(u[0])),
                            torch.sinh(0.2 * x[0]) - torch.cosh(0.1 * u[1]) + torch.erf(x[1])])

    def cost(x, u):
        return torch.atan2(x[0], 1.0 + x[1] ** 2) ** 2 + torch.arctan2(torch.sin(x[2]), torch.cos(x[2])) ** 2 + (u ** 2).sum()

    final = lambda x: torch.atan2(x[1], x[0]) ** 2
    src, params, info = fxsource.translate_ex(transition, cost, final, 3, 2)
    assert not info["cost_is_piecewise_linear"]
    rng = np.random.default_rng(2)
    for _ in range(4):
        x, u = rng.normal(size=3).astype(np.float32), rng.normal(size=2).astype(np.float32)
        assert np.allclose(_evaluate(src, "transition", params, x, u), transition(torch.as_tensor(x), torch.as_tensor(u)).numpy(), rtol=2e-5, atol=1e-6)
        assert np.isclose(_evaluate(src, "cost", params, x, u), float(cost(torch.as_tensor(x), torch.as_tensor(u))), rtol=2e-5)
        assert np.isclose(_evaluate(src, "final_cost", params, x, None), float(final(torch.as_tensor(x))), rtol=2e-5)


def _evaluate(source, name, p, x, u):
    """Runs the emitted statements of one function as Python (they are one assignment each, in C syntax that is also Python's but for the
    ternary, the float suffix and a few names)."""
    import math, re
    body = source[source.index(f" {name}(const float *p"):]
    body = body[body.index("{") + 1:body.index("\n}")]
    env = {"p": p, "x": x, "u": u, "x_next": [0.0] * len(x), "sqrt": math.sqrt, "sqrtf": math.sqrt, "exp": math.exp, "expf": math.exp,
           "log": math.log, "sin": math.sin, "cos": math.cos, "tanh": math.tanh, "abs": abs, "fabsf": abs, "tan": math.tan, "atan": math.atan, "asin": math.asin,
           "acos": math.acos, "sinh": math.sinh, "cosh": math.cosh, "erf": math.erf, "atan2": math.atan2, "atan2f": math.atan2, "max": lambda a, b: a if a >= b else b,
           "min": lambda a, b: a if a <= b else b, "fmaxf": max, "fminf": min, "pow": pow, "powf": pow, "S": float, "true": True, "false": False}
    env["tfmpc"] = type("ns", (), {"ad": type("ad", (), {"prim": staticmethod(float)})})
    result = None
    for line in body.strip().split("\n"):
        line = line.strip().rstrip(";")
        line = re.sub(r"(\d+\.\d+(?:e-?\d+)?)f", r"\1", line).replace("tfmpc::ad::prim", "float").replace("&&", " and ").replace("||", " or ")
        line = re.sub(r"\(!(\w+)\)", r"(not \1)", line)
        m = re.match(r"\((\w+) \? (.+) : (.+)\)$", line.split(" = ", 1)[1]) if " = " in line else None
        if line.startswith("return "):
            result = eval(line[7:], env)
        elif line.startswith("const "):
            _, _, rest = line.split(" ", 2)
            var, expr = rest.split(" = ", 1)
            if m:
                expr = f"({m.group(2)}) if {m.group(1)} else ({m.group(3)})"
            env[var] = np.float32(eval(expr, env)) if not isinstance(eval(expr, env), (bool, np.bool_)) else bool(eval(expr, env))
        else:
            var, expr = line.split(" = ", 1)
            exec(f"{var} = {expr}", env)
    return np.asarray(env["x_next"], dtype=np.float64) if name == "transition" else float(result)


@needs_hipcc
def test_the_generated_source_compiles_without_a_gpu():
    import ctypes
    env = torch_envs.reservoir(problems.RES4_CONFIG).to_device_env()
    path = deviceenv.build(env.source, 4, 4, env.n_zones)
    lib = ctypes.CDLL(path)
    assert lib.tfmpc_userenv_state_size() == 4 and lib.tfmpc_userenv_action_size() == 4


@needs_hipcc
@pytest.mark.parametrize("which,lanes", [("reservoir4", 16), ("hvac12", 32)])
def test_the_lane_packed_costate_kernel_is_in_the_companion_library(which, lanes):
    """Without a GPU: the companion library of a Python env whose cost is proved piecewise linear carries the row kernel instantiated for its size --
    sixteen lanes per instance up to n + m = 16, thirty-two up to 32 (csrc/user_env_group.h; the kernel's name is in the code object's metadata)."""
    n = int(which[9:]) if which.startswith("reservoir") else int(which[4:])
    cfg = (dict(problems.RES4_CONFIG) if n == 4 else dict(problems.reservoir_config(n, seed=5))) if which.startswith("reservoir") else problems.hvac_config(n, seed=5)
    env = (torch_envs.reservoir if which.startswith("reservoir") else torch_envs.hvac)(cfg).to_device_env()
    assert env.zero_cost_hessian
    path = deviceenv.build(env.source, n, n, env.n_zones, True)
    blob = open(path, "rb").read()
    assert (b"ilqr_user_costate_group_kernelILi%dELi%dELi%dEE" % (n, n, lanes)) in blob


# ---- on the device ---------------------------------------------------------------------------------------------------------------------------------

def _case(which):
    if which == "navigation":
        cfg = problems.NAV_CONFIG
        return cfg, Navigation.load(cfg), torch_envs.navigation(cfg, "cuda"), (0.0, 10.0), (-1.0, 1.0)
    if which.startswith("reservoir"):
        n = int(which[9:])
        cfg = dict(problems.RES4_CONFIG) if n == 4 else dict(problems.reservoir_config(n, seed=5))
        return cfg, Reservoir.load(dict(cfg)), torch_envs.reservoir(cfg, "cuda"), (20.0, 95.0), (0.0, 1.0)
    n = int(which[4:])
    cfg = problems.hvac_config(n, seed=5)
    return cfg, HVAC.load(cfg), torch_envs.hvac(cfg, "cuda"), (5.0, 35.0), (0.0, 1.0)


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("which", ["navigation", "reservoir4", "reservoir7", "hvac6"])
def test_all_thirteen_derivative_tensors_of_the_translated_env(which):
    cfg, builtin, python_env, xr, ur = _case(which)
    device_env = python_env.to_device_env()
    n, m = device_env.state_size, device_env.action_size
    rng = np.random.default_rng(3)
    B, T = 12, 9
    x = rng.uniform(*xr, size=(B, T + 1, n, 1)).astype(np.float32)
    u = rng.uniform(*ur, size=(B, T, m, 1)).astype(np.float32)
    got = iLQR(device_env).derivatives(x, u)
    ref = iLQR(builtin).derivatives(x, u)
    torch.cuda.synchronize()
    names = [f"{t}.{f}" for t, tup in zip("tcf", got) for f in tup._fields]
    for name, a, b in zip(names, [v for tup in got for v in tup], [v for tup in ref for v in tup]):
        a, b = _np(a), _np(b)
        assert a.shape == b.shape, name
        scale = max(np.abs(b).max(), 1.0)
        assert np.abs(a - b).max() <= 2e-5 * scale, (which, name, np.abs(a - b).max(), scale)
    # ... and torch.func autodiff of the very functions that were translated (what TorchEnv itself computes)
    tm = python_env.get_linear_transition(x[:, :-1], u)
    cm = python_env.get_quadratic_cost(x[:, :-1], u)
    for name, a, b in zip(names, list(got[0]) + list(got[1]), list(tm) + list(cm)):
        a, b = _np(a), _np(b).reshape(_np(a).shape)
        assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1.0), (which, name)
    # single steps through the DiffEnv protocol
    xn_u, xn_b = device_env.transition(x[:, 0], u[:, 0], batch=True), builtin.transition(x[:, 0], u[:, 0], batch=True)
    assert np.abs(_np(xn_u) - _np(xn_b)).max() <= 2e-5 * max(np.abs(_np(xn_b)).max(), 1.0)
    cu, cb = _np(device_env.cost(x[:, 0], u[:, 0], batch=True)), _np(builtin.cost(x[:, 0], u[:, 0], batch=True))
    assert np.abs(cu - cb).max() <= 1e-5 * np.abs(cb).max() + 1e-5


@pytest.mark.gpu
@needs_hipcc
def test_navigation_from_python_solves_like_the_builtin_kernel():
    cfg, builtin, python_env, _, _ = _case("navigation")
    device_env = python_env.to_device_env()
    rng = np.random.default_rng(4)
    B, T = 256, 50
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = np.stack([problems.scalar_uniform_actions(T, [-1, -1], [1, 1], rng) for _ in range(B)]).astype(np.float32)
    s_user, s_builtin = iLQR(device_env), iLQR(builtin)
    out = s_user.solve_device(x0, T, u_init=u0)
    ref = s_builtin.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    assert s_user.last_kernel.startswith("lane_group") and "user env" in s_user.last_kernel
    same = (out["iterations"] == ref["iterations"]).cpu().numpy()
    assert same.mean() >= 0.9, same.mean()
    cu, cb = _np(out["costs"]).sum(1), _np(ref["costs"]).sum(1)
    assert np.abs(cu - cb)[same].max() <= 2e-3 * np.abs(cb).max()
    assert np.median(np.abs(cu - cb) / np.abs(cb)) <= 1e-5
    assert int((out["status"] & ~_hip.ST_NOT_PD).abs().sum()) == 0
    # the host-driven TorchEnv solve of the same functions lands on the same costs (a handful of instances: it is the slow path)
    host = iLQR(python_env, compile_env=False)
    assert host.python_env is None and iLQR(python_env).python_env is python_env        # (the default translates; compile_env=False keeps the host loop)
    tj, _ = host.solve(x0[:8], T, show_progress=False, u_init=u0[:8])
    assert np.median(np.abs(tj.costs.sum(1) - cu[:8]) / np.abs(cu[:8])) <= 1e-3


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("which", ["reservoir4", "hvac6"])
def test_piecewise_linear_envs_from_python_take_the_same_first_iterations(which):
    """res4 / hvac6 (the reference's own configs) from Python: the cost's second derivatives are exactly zero through the translated max / abs
    too, so the backward pass takes the bang-bang branch; three iterations against the built-in wave kernel."""
    from oracle import envs_ref, ilqr_ref
    cfg, builtin, python_env, xr, _ = _case(which)
    device_env = python_env.to_device_env()
    n = device_env.state_size
    rng = np.random.default_rng(8)
    B, T = 24, 30
    x0 = rng.uniform(xr[0] + 0.3 * (xr[1] - xr[0]), xr[1] - 0.2 * (xr[1] - xr[0]), size=(B, n, 1)).astype(np.float32)
    u0 = iLQR(builtin).random_actions(T, B, seed=2)
    cm = iLQR(device_env).derivatives(iLQR(device_env).start(x0, T, u_init=u0)[0], u0)[1]
    for name in ("l_xx", "l_uu", "l_ux", "l_xu"):
        assert float(getattr(cm, name).abs().max()) == 0.0, name
    with _hip.option("TFMPC_ILQR_KERNEL", "wave"):
        ref = iLQR(builtin, max_iterations=3).solve_device(x0, T, u_init=u0)
    out = iLQR(device_env, max_iterations=3).solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    cu, cb = _np(out["costs"]).sum(1), _np(ref["costs"]).sum(1)
    assert int(out["status"].abs().sum()) == 0
    assert np.median(np.abs(cu - cb) / np.abs(cb)) <= 1e-3
    assert np.all((_np(out["actions"]) >= -1e-6) & (_np(out["actions"]) <= 1 + 1e-6))
    # ... and the FIRST iteration against the fp64 restatement of ilqr.py on the restated env (bang-bang iterates of two arithmetics part
    # ways after a few iterations -- a selector at a rounding-level Q_u -- so later iterations are compared with the built-in kernel only)
    oenv = (envs_ref.Reservoir if which.startswith("reservoir") else envs_ref.HVAC)(**cfg)
    one = iLQR(device_env, max_iterations=1).solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    # (without the final cost: at the LAST step Q_u = x_i (V_x[i+1] - V_x[i]) is an exact zero wherever two neighbouring reservoirs sit on the
    # same linear piece of the cost -- the selector `Q_u >= 0` then reads the sign of a rounding error, and the last action, hence x_T, is a tie)
    for b in (0, 5):
        xs, us, cs, it = ilqr_ref.ILQRRef(oenv, max_iterations=1).solve(x0[b].astype(np.float64), T, u_init=_np(torch.as_tensor(u0[b])))
        got = _np(one["costs"][b])[:-1].sum()
        assert abs(cs[:-1].sum() - got) <= 5e-3 * abs(cs[:-1].sum()), (b, cs[:-1].sum(), got)


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("which", ["reservoir4", "hvac6"])
def test_costate_form_of_a_user_env_equals_its_dense_form(which):
    """A user env whose cost the translator proved piecewise affine runs the COSTATE form of the fused kernel (one first-order dual evaluation per
    direction instead of second-order linearisation + dense backward pass; every step size of a line search at once, one per lane).  Against the very
    same device source compiled WITHOUT the promise (dense form: dual-number Hessians, which come out exactly zero, box-QP branch never taken): the
    reference's backward pass takes the bang-bang branch either way (ilqr.py:137-141), so decisions and trajectories agree like two fp32 programs."""
    from tfmpc.envs.deviceenv import DeviceEnv
    cfg, builtin, python_env, xr, _ = _case(which)
    fast = python_env.to_device_env()
    assert fast.zero_cost_hessian
    dense = DeviceEnv(fast.source, fast.state_size, fast.action_size, params=fast.params, low=0.0, high=1.0, zero_cost_hessian=False)
    n = fast.state_size
    rng = np.random.default_rng(11)
    B, T = 64, 40
    x0 = rng.uniform(xr[0] + 0.3 * (xr[1] - xr[0]), xr[1] - 0.2 * (xr[1] - xr[0]), size=(B, n, 1)).astype(np.float32)
    u0 = iLQR(builtin).random_actions(T, B, seed=3)
    a = iLQR(fast, max_iterations=6).solve_device(x0, T, u_init=u0, trace_rows=20)
    b = iLQR(dense, max_iterations=6).solve_device(x0, T, u_init=u0, trace_rows=20)
    torch.cuda.synchronize()
    assert int(a["status"].abs().sum()) == 0 and int(b["status"].abs().sum()) == 0
    # first pass: same nominal trajectory, same gradients up to the summation order -> same step size on (nearly) every instance, same cost
    first_alpha = (a["trace"][:, 0, 5] == b["trace"][:, 0, 5]).float().mean()
    assert float(first_alpha) >= 0.9
    ca, cb = _np(a["costs"]).sum(1), _np(b["costs"]).sum(1)
    assert np.median(np.abs(ca - cb) / np.abs(cb)) <= 1e-3
    assert np.all((_np(a["actions"]) >= -1e-6) & (_np(a["actions"]) <= 1 + 1e-6))


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("which,B,T", [("reservoir4", 203, 40), ("hvac6", 130, 33), ("reservoir7", 57, 25), ("reservoir4", 1, 12),
                                       ("hvac12", 75, 30), ("reservoir10", 41, 21), ("hvac16", 9, 17)])
def test_sixteen_lanes_per_instance_equal_the_wave_per_instance_kernel(which, B, T):
    """Round 6: a small user env on the costate path (n + m <= 16) runs SIXTEEN LANES per instance, four instances per wave (csrc/user_env_group.h) --
    the wave-per-instance kernel's two programs, one direction of z per lane in the backward pass and one step size per lane in the line search, in the
    16-lane rows of a wave.  The same arithmetic per instance, operation for operation: every output and the whole decision trace equal the
    wave-per-instance kernel's (`force_wave_kernel`) bit for bit; B is not a multiple of four, so the last wave carries idle rows.
    From n + m = 17 to 32 (hvac12, reservoir10, hvac16 = all 32 lanes) the same kernel runs THIRTY-TWO lanes per instance, two instances per wave."""
    cfg, builtin, python_env, xr, _ = _case(which)
    env = python_env.to_device_env()
    assert env.zero_cost_hessian
    n = env.state_size
    rng = np.random.default_rng(23)
    x0 = rng.uniform(xr[0] + 0.2 * (xr[1] - xr[0]), xr[1] - 0.1 * (xr[1] - xr[0]), size=(B, n, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=8)
    u0 = iLQR(builtin).random_actions(T, B, seed=4)
    outs = {}
    for wave in (False, True):
        env._library().force_wave_kernel(wave)
        try:
            o = solver.solve_device(x0, T, u_init=u0, trace_rows=24)
            torch.cuda.synchronize()
            outs[wave] = ({k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}, solver.last_kernel)
        finally:
            env._library().force_wave_kernel(False)
    assert outs[False][1].startswith("costate_group (%d lanes" % (16 if 2 * n <= 16 else 32)) and outs[True][1].startswith("wave")
    for key in ("states", "actions", "costs", "iterations", "status", "trace_len"):
        assert torch.equal(outs[False][0][key], outs[True][0][key]), key
    assert torch.equal(torch.nan_to_num(outs[False][0]["trace"], nan=-7.0), torch.nan_to_num(outs[True][0]["trace"], nan=-7.0))
    assert int(outs[False][0]["trace_len"].max()) >= 3 and (B < 40 or len(torch.unique(outs[False][0]["trace"][:, :3, 5])) >= 2)      # (the searches do backtrack)


@pytest.mark.gpu
@needs_hipcc
def test_the_bicycle_from_python_has_torch_funcs_derivatives_and_drives_to_its_goal():
    """A kinematic bicycle written in plain torch (tan, atan2 of sin / cos, ...: round 6's vocabulary): every derivative tensor of the compiled env
    against torch.func autodiff of the very functions that were translated (first AND second order: the nested dual numbers of the new
    functions), and a solve that reduces the cost inside the steering / acceleration limits."""
    python_env = torch_envs.bicycle("cuda")
    device_env = python_env.to_device_env()
    assert not device_env.zero_cost_hessian
    rng = np.random.default_rng(7)
    B, T = 10, 7
    x = np.concatenate([rng.uniform(-2, 9, size=(B, T + 1, 2, 1)), rng.uniform(-1.2, 1.2, size=(B, T + 1, 1, 1)), rng.uniform(0.2, 4, size=(B, T + 1, 1, 1))], axis=2).astype(np.float32)
    u = np.concatenate([rng.uniform(-2, 2, size=(B, T, 1, 1)), rng.uniform(-0.5, 0.5, size=(B, T, 1, 1))], axis=2).astype(np.float32)
    got = iLQR(device_env).derivatives(x, u)
    torch.cuda.synchronize()
    tm = python_env.get_linear_transition(x[:, :-1], u)
    cm = python_env.get_quadratic_cost(x[:, :-1], u)
    names = [f"{t}.{f}" for t, tup in zip("tc", got[:2]) for f in tup._fields]
    for name, a, b in zip(names, list(got[0]) + list(got[1]), list(tm) + list(cm)):
        a, b = _np(a), _np(b).reshape(_np(a).shape)
        assert np.abs(a - b).max() <= 5e-5 * max(np.abs(b).max(), 1.0), (name, np.abs(a - b).max(), np.abs(b).max())
    B, T = 128, 40
    x0 = np.concatenate([rng.uniform(-1, 1, size=(B, 2, 1)), rng.uniform(-0.5, 0.5, size=(B, 1, 1)), rng.uniform(0.5, 2, size=(B, 1, 1))], axis=1).astype(np.float32)
    u0 = np.zeros((B, T, 2, 1), dtype=np.float32)
    s = iLQR(device_env, max_iterations=40)
    start = _np(s.start(x0, T, u_init=u0)[2]).sum(1)
    out = s.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    cu = _np(out["costs"]).sum(1)
    assert int((out["status"] & ~_hip.ST_NOT_PD).abs().sum()) == 0
    assert np.all(cu <= start * (1 + 1e-6)) and np.median(cu / start) < 0.3
    act = _np(out["actions"])
    assert np.all(np.abs(act[:, :, 0]) <= 2.0 + 1e-6) and np.all(np.abs(act[:, :, 1]) <= 0.5 + 1e-6)


@pytest.mark.gpu
@needs_hipcc
def test_the_textbook_cart_pole_from_python():
    """The canonical iLQR example as three torch functions (n = 4, m = 1: the generic wave kernel): torch.func's derivatives on the device, and
    a solve from a tilted start with zero initial controls inside the force limit -- the cost falls on every instance and the pole ends UPRIGHT
    (the open-loop start lets it fall, so the local optimum iLQR finds from there brings it up the other way round: upright modulo a turn)."""
    python_env = torch_envs.cartpole("cuda")
    device_env = python_env.to_device_env()
    rng = np.random.default_rng(9)
    B, T = 10, 7
    x = rng.uniform(-1, 1, size=(B, T + 1, 4, 1)).astype(np.float32)
    u = rng.uniform(-10, 10, size=(B, T, 1, 1)).astype(np.float32)
    got = iLQR(device_env).derivatives(x, u)
    torch.cuda.synchronize()
    tm = python_env.get_linear_transition(x[:, :-1], u)
    cm = python_env.get_quadratic_cost(x[:, :-1], u)
    names = [f"{t}.{f}" for t, tup in zip("tc", got[:2]) for f in tup._fields]
    for name, a, b in zip(names, list(got[0]) + list(got[1]), list(tm) + list(cm)):
        a, b = _np(a), _np(b).reshape(_np(a).shape)
        assert np.abs(a - b).max() <= 5e-5 * max(np.abs(b).max(), 1.0), (name, np.abs(a - b).max(), np.abs(b).max())
    B, T = 128, 100
    x0 = np.concatenate([rng.uniform(-0.5, 0.5, size=(B, 2, 1)), rng.uniform(-0.3, 0.3, size=(B, 1, 1)), rng.uniform(-0.3, 0.3, size=(B, 1, 1))], axis=1).astype(np.float32)
    u0 = np.zeros((B, T, 1, 1), dtype=np.float32)
    s = iLQR(device_env, max_iterations=50)
    start = _np(s.start(x0, T, u_init=u0)[2]).sum(1)
    out = s.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    assert s.last_kernel.startswith("wave")
    cu = _np(out["costs"]).sum(1)
    assert int((out["status"] & ~_hip.ST_NOT_PD).abs().sum()) == 0
    assert np.all(cu <= start * (1 + 1e-6)) and np.median(cu / start) < 0.9
    assert np.all(np.abs(_np(out["actions"])) <= 10.0 + 1e-6)
    angle = _np(out["states"])[:, -1, 2, 0]
    assert np.median(np.abs(np.arctan2(np.sin(angle), np.cos(angle)))) < 0.15


@pytest.mark.gpu
@needs_hipcc
def test_a_pendulum_in_plain_torch_lands_on_the_lane_group_kernel():
    """An env of the user's own, written as three torch functions: traced, translated (sin included), compiled -- and, being dense with n + m <= 4,
    solved by the lane-group kernel (round 6); the same numbers as the hand-written device source of the same model."""
    import deviceenv_sources as sources
    from tfmpc.envs.deviceenv import DeviceEnv
    B, T = 256, 60
    rng = np.random.default_rng(5)
    x0 = np.stack([rng.uniform(-1.2, 1.2, size=B), rng.uniform(-1.0, 1.0, size=B)], axis=1).astype(np.float32)[..., None]
    u0 = np.zeros((B, T, 1, 1), dtype=np.float32)
    python_env = torch_envs.pendulum("cuda")
    s = iLQR(python_env, max_iterations=30)
    assert s.compile_error is None, s.compile_error
    out = s.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    assert s.last_kernel.startswith("lane_group"), s.last_kernel
    hand = DeviceEnv(sources.PENDULUM, 2, 1, params=np.array([0.05, 9.81, 0.1, 0.0, 1.0, 0.1, 0.01], dtype=np.float32), low=-4.0, high=4.0)
    ref = iLQR(hand, max_iterations=30).solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    cu, cr = _np(out["costs"]).sum(1), _np(ref["costs"]).sum(1)
    assert int(out["status"].abs().sum()) == 0
    assert (out["iterations"] == ref["iterations"]).float().mean() >= 0.9 and np.median(np.abs(cu - cr) / np.abs(cr)) <= 1e-4
    assert np.all(np.abs(_np(out["actions"])) <= 4.0 + 1e-6)


@pytest.mark.gpu
@needs_hipcc
def test_ilqr_takes_the_device_path_for_a_python_env_by_itself_and_falls_back_when_it_cannot():
    """`iLQR(TorchEnv(...))`: the functions are translated and compiled when possible -- the reference user changes nothing -- and an env the translator
    cannot take (here: torch.lgamma) runs the host-driven loop, with the reason kept."""
    cfg, builtin, python_env, _, _ = _case("navigation")
    solver = iLQR(python_env)
    assert solver.python_env is python_env and solver.compile_error is None and solver.env.kind == _hip.ENV_USER
    rng = np.random.default_rng(5)
    B, T = 64, 30
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = iLQR(builtin).random_actions(T, B, seed=2)
    traj, its = solver.solve(x0, T, show_progress=False, u_init=u0)
    ref, its_ref = iLQR(builtin).solve(x0, T, show_progress=False, u_init=u0)
    assert solver.last_kernel.startswith("lane_group") and (its == its_ref).mean() >= 0.9
    assert np.median(np.abs(traj.costs.sum(1) - ref.costs.sum(1)) / np.abs(ref.costs.sum(1))) <= 1e-5
    g = torch.as_tensor(np.array(cfg["goal"], dtype=np.float32).reshape(-1), device="cuda")
    odd = TorchEnv(lambda x, u: x + torch.lgamma(2.0 + u), lambda x, u: ((x - g) ** 2).sum(), lambda x: ((x - g) ** 2).sum(), 2, 2, -1.0, 1.0)
    fallback = iLQR(odd, max_iterations=3)
    assert fallback.python_env is None and isinstance(fallback.compile_error, fxsource.UnsupportedOperation) and fallback.env is odd
    out = fallback.solve_device(x0[:4], 10, u_init=u0[:4, :10])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out["costs"]).all())
    with pytest.raises(fxsource.UnsupportedOperation):
        iLQR(odd, compile_env=True)
