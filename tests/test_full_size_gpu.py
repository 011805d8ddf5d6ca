"""BASELINE.json configurations at FULL size that are not bench lines (cfg2, cfg5):
size-independent properties on every instance plus oracle parity on a sample.  (cfg3
and cfg4 at full size live in test_lqr_gpu.py / test_ilqr_gpu.py.)"""

import numpy as np
import pytest
import torch

import problems
from oracle import envs_ref, ilqr_ref, lqr_ref
from tfmpc import _hip
from tfmpc.envs import make_lqr_linear_navigation
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

pytestmark = pytest.mark.gpu


def test_cfg2_navlin_batch_4096():
    """navlin 2-D linear navigation, horizon 50, batch 4096 independent (init, goal) pairs:
    F, C shared (stride 0), goal-dependent c and x0 per instance.  Tolerance 1e-5 relative
    (BASELINE.json) against the fp64 oracle on a sample; invariants on all."""
    B, T, beta = 4096, 50, 5.0
    F, f, C, c, x0, goal = problems.make_navlin_batch(B, beta)
    lqr = make_lqr_linear_navigation(goal[..., None], beta)
    out = lqr.solve_device(x0[..., None], T, want_policy=True)
    torch.cuda.synchronize()
    assert int(out["status"].abs().sum()) == 0
    states, actions, costs = out["states"][..., 0], out["actions"][..., 0], out["costs"][:, :, 0, 0]
    assert torch.allclose(states[:, 1:], states[:, :-1] + actions, atol=1e-5)             # x' = x + u
    # shared F, C => the feedback gain K_t is the same for every instance
    assert torch.equal(out["K"][0], out["K"][B - 1])
    g = torch.as_tensor(goal, device=states.device, dtype=torch.float32)
    assert float((states[:, -1] - g).abs().max()) < 0.5                                   # reaches the goal
    for b in np.linspace(0, B - 1, 24).astype(int):
        x, u, cs, _, _ = lqr_ref.solve(F, f, C, c[b], x0[b], T)
        # actions are differences of states, so their absolute accuracy is set by the state scale
        for got, ref, scale in ((states[b], x, np.abs(x).max()), (actions[b], u, np.abs(x).max()),
                                (costs[b], cs, np.abs(cs).max())):
            assert np.abs(got.cpu().numpy() - ref).max() <= 1e-5 * scale


@pytest.mark.parametrize("kind", ["hvac", "reservoir"])
def test_cfg5_n32_horizon100_batch_32768(kind):
    """iLQR on HVAC / Reservoir at n = m = 32 (the reference envs have action_size ==
    state_size), horizon 100, batch 32 768, parameters from the reference's
    tests/conftest.py recipes.  Piecewise-linear costs => V_xx == 0, bang-bang backward (F6)."""
    n, T, B = 32, 100, 32768
    rng = np.random.default_rng(5)
    if kind == "hvac":
        cfg = problems.hvac_config(n, seed=5)
        env, oenv = HVAC.load(dict(cfg)), envs_ref.HVAC(**cfg, dtype=np.float32)
        x0 = np.full((B, n, 1), 10.0, dtype=np.float32) + rng.normal(0, 1.0, size=(B, n, 1)).astype(np.float32)
    else:
        cfg = problems.reservoir_config(n, seed=5)
        env, oenv = Reservoir.load(dict(cfg)), envs_ref.Reservoir(**cfg, dtype=np.float32)
        x0 = rng.uniform(50.0, 75.0, size=(B, n, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=12)         # bounded wall time: the property holds for any budget
    u0 = solver.random_actions(T, B, seed=5)
    out = solver.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    states, actions, costs = out["states"][..., 0], out["actions"][..., 0], out["costs"]
    assert torch.isfinite(states).all() and torch.isfinite(costs).all()
    assert int((out["status"] & (_hip.ST_NAN | _hip.ST_NOT_PD)).sum()) == 0
    assert float(actions.min()) >= 0.0 and float(actions.max()) <= 1.0                     # box [0, 1]
    start_cost = solver.start(x0, T, u_init=u0)[2].sum(dim=1)
    assert bool((costs.sum(dim=1) <= start_cost * (1 + 1e-5) + 1e-3).all())               # monotone improvement
    idx = torch.arange(0, B, 512, device=states.device)
    for t in (0, 49, 99):                                                                  # trajectory obeys the env
        nxt = env.transition(states[idx, t].unsqueeze(-1), actions[idx, t].unsqueeze(-1), batch=True)
        if kind == "reservoir":       # the 16-per-wave solve kernel keeps the env kernel's operation order here
            assert torch.equal(nxt[..., 0], states[idx, t + 1])
        else:                         # HVAC: the solve kernel folds the linear terms into the conduction matrix
            assert float((nxt[..., 0] - states[idx, t + 1]).abs().max()) <= 1e-6 * float(states[idx, t + 1].abs().max())
    # sample parity with the fp32 restatement on the first backward/forward pieces (deterministic part)
    b = 7
    o = ilqr_ref.ILQRRef(oenv, dtype=np.float32)
    xs, us, cs = o.start(x0[b], T, u_init=u0[b].cpu().numpy())
    gs = solver.start(x0[b], T, u_init=u0[b])
    assert np.abs(gs[0].cpu().numpy() - xs).max() <= 2e-4 * np.abs(xs).max()
    models = o.derivatives(xs, us)
    K, k, J, dV1, dV2 = o.backward(T, us, *models, mu=0.0)
    Kg, kg, Jg, d1, d2 = solver.backward(T, us, *solver.derivatives(xs, us), mu=0.0)
    assert not K.any() and not bool(Kg.any())                                              # F6: K == 0
    assert np.array_equal(kg.cpu().numpy() == (0.0 - us), k == (0.0 - us))                  # same bang-bang pattern
    assert abs(float(Jg) - float(J)) <= 1e-4 * abs(float(J))


def test_bf16_storage_mode_is_a_bounded_perturbation_on_hvac():
    """BASELINE configs[4] "fp32 vs bf16": rounding stored trajectories / gains to bf16 (fp32
    arithmetic) must perturb, not break, the solve (full sweep: tests/bf16_sweep.py)."""
    n, T, B = 32, 100, 256
    env = HVAC.load(dict(problems.hvac_config(n, seed=5)))
    x0 = (10.0 + np.random.default_rng(5).normal(0, 1.0, size=(B, n, 1))).astype(np.float32)
    u0 = iLQR(env).random_actions(T, B, seed=5)
    outs = {}
    for mode in (False, True):
        outs[mode] = iLQR(env, max_iterations=1, storage_bf16=mode).solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    a, b = outs[False]["states"], outs[True]["states"]
    rel = ((a - b).abs().reshape(B, -1).amax(1) / a.abs().reshape(B, -1).amax(1)).cpu().numpy()
    assert 1e-4 < np.median(rel) < 5e-2 and rel.max() < 0.2        # bf16 has 8 significant bits
    # stored values really are bf16-representable
    bits = b.contiguous().view(torch.int32)
    assert int((bits & 0xFFFF).abs().sum()) == 0


def test_cfg5_literal_dims_as_ilqr_on_the_lq_env():
    """BASELINE configs[4] names n = 32, m = 16, horizon 100, batch 32 768 -- dims no reference env has (all of them have
    action_size == state_size).  The survey allows a generalised env: the LQ env (lqr.py:36-57 through the
    DiffEnv protocol) admits n != m, and iLQR on it runs the dense regularised backward pass (ilqr.py:94-172, Cholesky
    controller) at that shape.  Oracle parity on a sample, size-independent properties on the whole batch."""
    from tfmpc.envs.lq import LQEnv
    from tfmpc.solvers.lqr import LQR
    n, m, T, B = 32, 16, 100, 32768
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=6)
    F = F * (0.9 / np.sqrt(n))                       # spectral radius ~0.9: the open-loop start rollout stays in fp32 range over 100 steps
    solver = iLQR(LQEnv(F, f, C, c))
    x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
    u0 = torch.zeros(B, T, m, 1, device="cuda")
    out = solver.solve_device(x0d, T, u_init=u0)
    torch.cuda.synchronize()
    assert int(out["status"].abs().sum()) == 0
    its = out["iterations"].cpu().numpy()
    assert its.mean() <= 1.2 and its.max() <= 4       # a convex LQ problem: one Newton step and its confirmation (a few need a shorter step first)
    states, actions, costs = out["states"][..., 0], out["actions"][..., 0], out["costs"]
    assert torch.isfinite(states).all() and torch.isfinite(costs).all()
    z = torch.cat([states[:, :-1], actions], dim=-1).double()
    Fd = torch.as_tensor(F, device="cuda", dtype=torch.float64)
    fd = torch.as_tensor(f, device="cuda", dtype=torch.float64)
    pred = torch.einsum("bij,btj->bti", Fd, z) + fd[:, None, :]
    rel = (pred - states[:, 1:].double()).abs().amax(dim=(1, 2)) / states.abs().amax(dim=(1, 2)).double().clamp_min(1.0)
    assert float(rel.max()) < 2e-5
    # iLQR lands on the LQR optimum of the same problem (sample)
    idx = np.linspace(0, B - 1, 64).astype(int)
    lq = LQR(F[idx], f[idx], C[idx], c[idx]).solve_device(x0[idx], T)
    tot_i, tot_l = costs[torch.as_tensor(idx, device="cuda")].sum(dim=1), lq["costs"][:, :, 0, 0].sum(dim=1)
    assert float(((tot_i - tot_l).abs() / lq["costs"].abs().sum(dim=(1, 2, 3))).max()) <= 2e-3
    # fp64 oracle (and the fp32 restatement for the budget) on one instance (TFMPC_SLOW=1: two -- each costs ~10 s of restatement at this shape)
    import os
    for b in ((0, B - 1) if os.environ.get("TFMPC_SLOW") == "1" else (B - 1,)):
        o = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b]))
        x, u, cs, it = o.solve(x0[b], T, u_init=np.zeros((T, m, 1)))
        o32 = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b], dtype=np.float32), dtype=np.float32)
        x32, u32, c32, _ = o32.solve(x0[b].astype(np.float32), T, u_init=np.zeros((T, m, 1), dtype=np.float32))
        assert it == int(its[b])
        for got, r64, r32, what in ((states[b], x, x32, "states"), (actions[b], u, u32, "actions"), (costs[b], cs, c32, "costs")):
            allowed = 5 * max(np.abs(r32.astype(np.float64) - r64).max(), 1e-5 * np.abs(r64).max())
            err = np.abs(got.cpu().numpy().astype(np.float64) - r64).max()
            assert err <= allowed, (b, what, err, allowed)
