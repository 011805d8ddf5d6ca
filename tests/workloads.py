"""The secondary iLQR workloads of ``bench.py`` (``extra.ilqr_api`` and friends), defined ONCE so that the bench line and the
decision-trace parity tests (tests/test_ilqr_lq_trace_gpu.py) run the very same problems.  Needs the GPU (start actions of
the warm-start workload come from ``LQR.solve`` on the device and a device-side noise generator); all returned arrays
are numpy float64 / float32 on the host plus the device tensors a solver call takes.

Every workload carries a ``version`` tag that ``bench.py`` prints with its numbers: a number is comparable across
rounds only under the same tag (the definitions changed in round 3 without one)."""

import numpy as np
import torch

import problems

N, M, T = 16, 8, 50


def ilqr_api_warm(B, n=N, m=M, horizon=T):
    """``iLQR.solve`` on ``LQEnv(0.25 F, f, C, c)`` of the reference's ``make_lqr`` distribution
    (tests/problems.py:make_lqr_batch_spd), start = the LQR-optimal open-loop actions + 5 % noise (an MPC re-solve;
    ~2 iterations, the first step size accepted).  Round 3's definition, unchanged."""
    from tfmpc.solvers.lqr import LQR
    F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=4321)
    F = 0.25 * F
    x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
    opt = LQR(F, f, C, c).solve_device(x0d, horizon)["actions"]
    gen = torch.Generator(device="cuda").manual_seed(7)
    u0 = (opt + 0.05 * opt.abs().amax(dim=(1, 2, 3), keepdim=True) * torch.randn(opt.shape, device="cuda", generator=gen)).contiguous()
    return dict(version="ilqr_api/warm-r3", F=F, f=f, C=C, c=c, x0=x0d, u0=u0, low=None, high=None, T=horizon,
                text=f"iLQR.solve on LQEnv(0.25 F, f, C, c) of make_lqr's distribution, n={n} m={m} T={horizon} B={B}, "
                     "start = LQR-optimal actions + 5 % noise")


def ilqr_api_cold(B, n=N, m=M, horizon=T):
    """The same problems from a COLD start: zero actions (the line search has to backtrack on part of the batch and the
    solve takes more iterations than from the warm start)."""
    F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=4321)
    F = 0.25 * F
    x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
    u0 = torch.zeros(B, horizon, m, 1, device="cuda")
    return dict(version="ilqr_api/cold-r4", F=F, f=f, C=C, c=c, x0=x0d, u0=u0, low=None, high=None, T=horizon,
                text=f"the same LQEnv problems, n={n} m={m} T={horizon} B={B}, start = zero actions")


def control_limited(B, n=N, m=M, horizon=T, bound=0.5):
    """``LQEnv(0.25 F, f, C, c)`` of the well-conditioned generator (tests/problems.py:make_lqr_batch_fast, eigenvalues of
    C in [1, 2]), zero start, actions in [-bound, bound]: the box-QP at every backward step (ilqr.py:136-138, 364-387).
    Round 3's definition, unchanged (rounds 1-2 used the same generator)."""
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=4321)
    F = 0.25 * F
    x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
    u0 = torch.zeros(B, horizon, m, 1, device="cuda")
    return dict(version="control_limited/fast-r3", F=F, f=f, C=C, c=c, x0=x0d, u0=u0, low=-bound, high=bound, T=horizon,
                text=f"LQEnv(0.25 F, f, C, c) of the well-conditioned generator, zero start, actions in [{-bound}, {bound}] "
                     f"(ilqr.py:136-138,364-387: box-QP at every step), B={B}")


def control_limited_stable(B, n=N, m=M, horizon=T, bound=0.5):
    """The same control-limited problems with F scaled to spectral radius ~0.7 (0.18 F): the zero-action open-loop start stays
    bounded, so fp32 can pose every instance.  With 0.25 F (``control_limited``) ~13 % of the batch starts from a rollout whose cost
    is 1e12 .. 1e21: there fp32 rounding of V_xx makes Q~_uu indefinite until mu ~ 1e4 .. 1e13 drowns it, every fp32 program (this
    kernel and the fp32 restatement alike, same regularisation level on 19 of 22 sampled) crawls for up to 100 iterations, and the
    fp64 restatement -- no Cholesky failure at all -- solves the instance in 30-60 (profiles/r04_box_family_oracle.json).  That tail
    is most of `control_limited`'s launch time; this workload is the one without it."""
    w = control_limited(B, n, m, horizon, bound)
    w["F"] = w["F"] * (0.18 / 0.25)
    w["version"] = "control_limited/stable-r4"
    w["text"] = (f"LQEnv(0.18 F, f, C, c) of the well-conditioned generator (spectral radius ~0.7), zero start, actions in "
                 f"[{-bound}, {bound}], B={B}")
    return w


def literal_dims(B, n=32, m=16, horizon=100):
    """BASELINE configs[4] at its literal dims as iLQR on the LQ env: well-conditioned generator, F scaled to
    spectral radius ~0.9, zero start."""
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=6)
    F = F * (0.9 / np.sqrt(n))
    x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
    u0 = torch.zeros(B, horizon, m, 1, device="cuda")
    return dict(version="literal_dims/fast-r2", F=F, f=f, C=C, c=c, x0=x0d, u0=u0, low=None, high=None, T=horizon,
                text=f"iLQR.solve on LQEnv(0.9/sqrt(n) F, f, C, c), n={n} m={m} T={horizon} B={B}, zero start")


def cfg5(kind, B, n=32, horizon=100):
    """BASELINE configs[4] on the reference's HVAC / Reservoir envs at n = m = 32, T = 100: parameters from the recipes
    of the reference's tests/conftest.py:35-63 / :83-127 (tests/problems.py), one env shared by the batch; x0 = 10.0 (HVAC,
    hvac6.config.json:32) or U(50, 75) per reservoir (res4.config.json:20), start actions = one scalar uniform per step (ilqr.py:70).
    `bench.py` solves it with ``iLQR(env, max_iterations=12)``."""
    from tfmpc.envs.hvac import HVAC
    from tfmpc.envs.reservoir import Reservoir
    from tfmpc.solvers.ilqr import iLQR
    if kind == "hvac":
        cfg = dict(problems.hvac_config(n, seed=5))
        env, x0 = HVAC.load(dict(cfg)), np.full((B, n, 1), 10.0, dtype=np.float32)
    else:
        cfg = dict(problems.reservoir_config(n, seed=5))
        env, x0 = Reservoir.load(dict(cfg)), np.random.default_rng(55).uniform(50, 75, size=(B, n, 1)).astype(np.float32)
    u0 = iLQR(env).random_actions(horizon, B, seed=5)
    return dict(version=f"cfg5/{kind}-r5", kind=kind, cfg=cfg, env=env, x0=torch.as_tensor(x0, device="cuda"), u0=u0, T=horizon,
                text=f"{kind} n=m={n} T={horizon} B={B}, shared env, <= 12 iterations")


def solver_of(w, **kwargs):
    from tfmpc.envs.lq import LQEnv
    from tfmpc.solvers.ilqr import iLQR
    return iLQR(LQEnv(w["F"], w["f"], w["C"], w["c"], low=w["low"], high=w["high"]), **kwargs)


def instance_cfg(w, b, u0_host=None):
    """One instance of a workload as the per-instance config ``tests/trace_oracle.py`` takes (kind ``"lq"``)."""
    return dict(F=w["F"][b], f=w["f"][b], C=w["C"][b], c=w["c"][b], low=w["low"], high=w["high"])


# ---- the other lines of bench.py's `other_configs` (round 6: defined here too, so that tools/profile_workload.py -- the process rocprofv3
# runs -- launches the very problems bench.py times; until round 5 they were built inline in bench.py) ------------------------------------------

def headline(B, n=N, m=M, seed=1234):
    """The headline LQR batch of rank 0 (bench.py main: tests/problems.py:make_lqr_batch_spd, seed 1234 + rank)."""
    from tfmpc.solvers.lqr import LQR
    F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=seed)
    return dict(version="headline/spd-r2", lqr=LQR(F, f, C, c), x0=torch.as_tensor(x0[..., None].astype(np.float32), device="cuda"), T=T)


def cfg2():
    """BASELINE configs[1]: navlin LQR, beta = 5, B = 4 096, T = 50 (tests/problems.py:make_navlin_batch)."""
    from tfmpc.envs import make_lqr_linear_navigation
    F, f, C, c, x0n, goal = problems.make_navlin_batch(4096, 5.0)
    lqr = make_lqr_linear_navigation(goal[..., None], 5.0)
    return dict(version="cfg2/navlin-r1", lqr=lqr, x0=torch.as_tensor(np.ascontiguousarray(x0n[..., None], dtype=np.float32), device="cuda"), T=50)


def cfg4(batches=1):
    """BASELINE configs[3]: Navigation iLQR, n = m = 2, T = 50; `batches` = 1: the config's 16 384 instances (x0 ~ U(0, 10)^2 from
    default_rng(4), start actions seed 4); 8: ONE launch of 8 x 16 384 instances (seeds 100 .. 107)."""
    from tfmpc.envs.navigation import Navigation
    from tfmpc.solvers.ilqr import iLQR
    solver = iLQR(Navigation.load(problems.NAV_CONFIG))
    Bn = 16384
    if batches == 1:
        x0 = torch.as_tensor(np.random.default_rng(4).uniform(0, 10, size=(Bn, 2, 1)).astype(np.float32), device="cuda")
        u0 = solver.random_actions(50, Bn, seed=4)
    else:
        x0 = torch.as_tensor(np.concatenate([np.random.default_rng(100 + i).uniform(0, 10, size=(Bn, 2, 1)) for i in range(batches)]).astype(np.float32), device="cuda")
        u0 = torch.cat([solver.random_actions(50, Bn, seed=100 + i) for i in range(batches)])
    return dict(version=f"cfg4/nav-r2x{batches}", solver=solver, x0=x0, u0=u0, T=50)


def small_env(name, B=16384, horizon=100):
    """The reference's own env configs (hvac6.config.json, res4.config.json) at B = 16 384, T = 100, <= 12 iterations: the config's initial
    state times U(0.9, 1.1) per instance -- drawn from default_rng(4) AFTER cfg4's x0 (and, for res4, after hvac6's factors): the order
    bench.py drew them in when these lines were built inline (rounds 2-5), kept so that the numbers stay comparable."""
    from tfmpc.envs.hvac import HVAC
    from tfmpc.envs.reservoir import Reservoir
    from tfmpc.solvers.ilqr import iLQR
    rng = np.random.default_rng(4)
    rng.uniform(0, 10, size=(16384, 2, 1))
    factors = {"hvac6": rng.uniform(0.9, 1.1, size=(B, 1, 1))}
    factors["res4"] = rng.uniform(0.9, 1.1, size=(B, 1, 1))
    env, x0r = (HVAC.load(dict(problems.HVAC6_CONFIG)), problems.HVAC6_X0) if name == "hvac6" else (Reservoir.load(dict(problems.RES4_CONFIG)), problems.RES4_X0)
    x0 = torch.as_tensor((np.array(x0r, dtype=np.float32)[None] * factors[name]).astype(np.float32), device="cuda")
    solver = iLQR(env, max_iterations=12)
    return dict(version=f"small_env/{name}-r2", solver=solver, x0=x0, u0=solver.random_actions(horizon, B, seed=1), T=horizon, n=len(x0r))


def lqr32():
    """A dense LQR beyond the headline tile: n = 32, m = 16, T = 50, B = 8 192 (make_lqr_batch_fast seed 1, 0.5 F)."""
    from tfmpc.solvers.lqr import LQR
    F, f, C, c, x0 = problems.make_lqr_batch_fast(8192, 32, 16, seed=1)
    lqr = LQR(0.5 * F, f, C, c)
    return dict(version="lqr32/fast-r2", lqr=lqr, x0=lqr._prep_x0(x0), T=50)
