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
