"""Matrix-core iLQR solve for the LQ env (n <= 16, m <= 8, unbounded): the BASELINE headline
shape driven through the iLQR API.  Checked against the generic wave kernel, the fp64 oracle and
the LQR optimum, including the second-chance path for instances that need regularisation."""

import os

import numpy as np
import pytest
import torch

import problems
from oracle import envs_ref, ilqr_ref
from tfmpc import _hip
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
from tfmpc.solvers.lqr import LQR

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_kernel():
    def set_(name):
        _hip.set_option("TFMPC_ILQR_KERNEL", name)
    yield set_
    set_(None)


def _problem(B, n, m, seed, scale=0.25):
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=seed)
    return F * scale * np.sqrt(16.0 / n), f, C, c, x0.astype(np.float32)


@pytest.mark.parametrize("n,m,T", [(16, 8, 50), (16, 8, 7), (12, 6, 20), (9, 3, 16), (16, 1, 10)])
def test_mfma_solve_matches_wave_kernel_oracle_and_lqr(force_kernel, n, m, T):
    B = 80
    F, f, C, c, x0 = _problem(B, n, m, seed=100 * n + m)
    solver = iLQR(LQEnv(F, f, C, c))
    u0 = (0.1 * np.random.default_rng(1).normal(size=(B, T, m, 1))).astype(np.float32)
    out = {}
    for kern in (None, "wave"):
        force_kernel(kern)
        out[kern] = solver.solve_device(x0[..., None], T, u_init=u0)
        torch.cuda.synchronize()
        assert int(out[kern]["status"].abs().sum()) == 0, kern
    mf, wv = out[None], out["wave"]
    assert torch.equal(mf["iterations"], wv["iterations"])
    # two fp32 programs with different summation order: they agree to ~1e-5 on the typical
    # instance, while a few ill-conditioned ones sit 1e-2 from fp64 in EVERY fp32 implementation
    # (fp32 restatement included), so the bar is on the distribution over the batch
    for key in ("states", "actions", "costs"):
        diff = (mf[key] - wv[key]).abs().reshape(B, -1).amax(dim=1)
        scale = wv[key].abs().reshape(B, -1).amax(dim=1).clamp_min(1e-6)
        rel = (diff / scale).cpu().numpy()
        assert np.median(rel) <= 1e-4 and np.quantile(rel, 0.9) <= 5e-3 and rel.max() <= 5e-2, (key, np.median(rel), rel.max())
    # iLQR on an LQ problem lands on the LQR optimum (one Newton step + confirmation)
    assert int(mf["iterations"].max()) <= 2
    lq = LQR(F, f, C, c).solve_device(x0, T)
    tot_i, tot_l = mf["costs"].sum(dim=1), lq["costs"][:, :, 0, 0].sum(dim=1)
    assert float(((tot_i - tot_l).abs() / lq["costs"].abs().sum(dim=(1, 2, 3))).max()) <= 2e-3
    # fp64 oracle on two instances
    for b in (0, B - 1):
        o = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b]))
        x, u, cs, it = o.solve(x0[b], T, u_init=u0[b])
        assert it == int(mf["iterations"][b])
        o32 = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b], dtype=np.float32), dtype=np.float32)
        x32, _, _, _ = o32.solve(x0[b], T, u_init=u0[b])
        budget = 5 * max(np.abs(x32 - x).max(), 1e-5 * np.abs(x).max())
        assert np.abs(mf["states"][b, ..., 0].cpu().numpy() - x).max() <= budget
        assert abs(float(tot_i[b]) - cs.sum()) <= 2e-3 * np.abs(cs).sum()


def test_second_chance_launch_handles_non_pd_instances(force_kernel):
    """Instances whose Q_uu is not positive definite (negative C_uu here) need mu > 0, which the
    matrix-core kernel does not implement: it flags them and the wave kernel re-solves exactly
    those, so the result equals an all-wave run bit for bit; healthy instances stay on the fast path."""
    B, n, m, T = 48, 16, 8, 10
    F, f, C, c, x0 = _problem(B, n, m, seed=9)
    bad = np.arange(B) % 5 == 0
    C[bad, n:, n:] = -0.5 * np.eye(m)                 # concave in u at the last step
    solver = iLQR(LQEnv(F, f, C, c), max_iterations=6, max_attempts=8)
    u0 = np.zeros((B, T, m, 1), dtype=np.float32)
    force_kernel(None)
    mixed = solver.solve_device(x0[..., None], T, u_init=u0)
    force_kernel("wave")
    wave = solver.solve_device(x0[..., None], T, u_init=u0)
    torch.cuda.synchronize()
    badt = torch.as_tensor(bad, device="cuda")
    assert int((mixed["status"] & 0x4000).sum()) == 0                 # the internal bit never leaks
    assert bool(((mixed["status"][badt] & _hip.ST_NOT_PD) != 0).all())
    for key in ("states", "actions", "costs", "iterations", "status"):
        assert torch.equal(mixed[key][badt], wave[key][badt]), key
    good = ~badt
    assert int(mixed["status"][good].abs().sum()) == 0
    assert float((mixed["states"][good] - wave["states"][good]).abs().max()) <= 5e-4 * float(wave["states"][good].abs().max())


def test_bounded_lq_env_is_clipped():
    """(bounded LQ envs run on the control-limited matrix-core kernel: tests/test_ilqr_lq_box_mfma_gpu.py)"""
    lib = _hip.require_gpu()
    F, f, C, c, x0 = _problem(4, 16, 8, seed=3)
    solver = iLQR(LQEnv(F, f, C, c, low=-0.5, high=0.5))
    out = solver.solve_device(x0[..., None], 8, u_init=np.zeros((4, 8, 8, 1), dtype=np.float32))
    torch.cuda.synchronize()
    assert float(out["actions"].abs().max()) <= 0.5 + 1e-6           # clipped => box-QP path ran


@pytest.mark.parametrize("n,m,T", [(32, 16, 100), (32, 16, 7), (24, 12, 20), (17, 9, 30), (32, 3, 16), (20, 16, 55), (9, 16, 12)])
def test_large_tile_solve_matches_wave_kernel_oracle_and_lqr(force_kernel, n, m, T):
    """Unbounded LQ env beyond the 16 x 8 tile, up to n = 32, m = 16 (BASELINE configs[4]'s literal dims): the 2 x 2-tile
    matrix-core solve (ilqr_lq_mfma32.hip, trajectories in HBM) against the wave kernel, the LQR optimum and the fp64 /
    fp32 restatement of ilqr.py."""
    B = 48
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=100 * n + m)
    F = F * (0.9 / np.sqrt(n))
    x0 = x0.astype(np.float32)
    solver = iLQR(LQEnv(F, f, C, c))
    u0 = (0.1 * np.random.default_rng(1).normal(size=(B, T, m, 1))).astype(np.float32)
    out = {}
    for kern in (None, "wave"):
        force_kernel(kern)
        out[kern] = solver.solve_device(x0[..., None], T, u_init=u0)
        torch.cuda.synchronize()
        assert int(out[kern]["status"].abs().sum()) == 0, kern
    mf, wv = out[None], out["wave"]
    assert float((mf["iterations"] == wv["iterations"]).float().mean()) >= 0.9
    for key in ("states", "actions", "costs"):
        diff = (mf[key] - wv[key]).abs().reshape(B, -1).amax(dim=1)
        scale = wv[key].abs().reshape(B, -1).amax(dim=1).clamp_min(1e-6)
        rel = (diff / scale).cpu().numpy()
        assert np.median(rel) <= 1e-4 and np.quantile(rel, 0.9) <= 5e-3 and rel.max() <= 5e-2, (key, np.median(rel), rel.max())
    lq = LQR(F, f, C, c).solve_device(x0, T)
    tot_i, tot_l = mf["costs"].sum(dim=1), lq["costs"][:, :, 0, 0].sum(dim=1)
    assert float(((tot_i - tot_l).abs() / lq["costs"].abs().sum(dim=(1, 2, 3))).max()) <= 2e-3
    for b in (0, B - 1):
        o = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b]))
        x, u, cs, it = o.solve(x0[b], T, u_init=u0[b])
        assert it == int(mf["iterations"][b])
        o32 = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b], dtype=np.float32), dtype=np.float32)
        x32, _, _, _ = o32.solve(x0[b], T, u_init=u0[b])
        budget = 5 * max(np.abs(x32 - x).max(), 1e-5 * np.abs(x).max())
        assert np.abs(mf["states"][b, ..., 0].cpu().numpy() - x).max() <= budget
        assert abs(float(tot_i[b]) - cs.sum()) <= 2e-3 * np.abs(cs).sum()


@pytest.mark.parametrize("n,m,T", [(16, 8, 50), (12, 6, 20), (9, 3, 16), (32, 16, 40), (24, 12, 20), (20, 5, 16)])
def test_gain_reusing_later_passes_agree_with_full_passes(n, m, T):
    """Round 6 (TFMPC_ILQR_LQ_REUSE): for a time-invariant LQ env at mu = 0 the matrices of the backward pass (Q_xx, Q_ux, Q_uu -> K_t, V_xx)
    do not depend on the trajectory, so from the second pass on the kernel keeps K_t and Q_uu^-1 of the first and runs the vector recursion
    (ilqr.py:122-123,152-156) alone.  Against the same kernel with the full pass in every iteration (=0, what ilqr.py:94-172 does): the same
    decisions, and numbers that agree like two fp32 programs with different summation order.  An unreachable atol makes every instance run all
    its iterations, i.e. five gain-reusing passes in a row, each followed by a line search on ITS k_t."""
    B = 96
    if n > 16:                       # the large-tile twin (ilqr_lq_mfma32.hip: -Q_uu^-1 through fifteen identity columns + the last pivot's reciprocal)
        F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=7 * n + m)
        F, x0 = F * (0.9 / np.sqrt(n)), x0.astype(np.float32)
    else:
        F, f, C, c, x0 = _problem(B, n, m, seed=7 * n + m)
    u0 = (0.1 * np.random.default_rng(2).normal(size=(B, T, m, 1))).astype(np.float32)
    for kwargs, min_passes in ((dict(), 2), (dict(atol=1e-12, max_iterations=6), 6)):
        solver = iLQR(LQEnv(F, f, C, c), **kwargs)
        rows = 8
        out = {}
        for mode in (None, "0"):
            with _hip.option("TFMPC_ILQR_LQ_REUSE", mode):
                out[mode] = solver.solve_device(x0[..., None], T, u_init=u0, trace_rows=rows)
            torch.cuda.synchronize()
            assert int(out[mode]["status"].abs().sum()) == 0
            assert solver.last_kernel.startswith("lq_mfma")
        re, full = out[None], out["0"]
        assert float((re["iterations"] == full["iterations"]).float().mean()) >= 0.97
        assert int(re["trace_len"].min()) >= min_passes
        same = (re["iterations"] == full["iterations"])
        for key in ("states", "actions", "costs"):
            diff = (re[key] - full[key]).abs().reshape(B, -1).amax(dim=1)
            scale = full[key].abs().reshape(B, -1).amax(dim=1).clamp_min(1e-6)
            rel = (diff / scale)[same].cpu().numpy()
            assert np.median(rel) <= 1e-5 and rel.max() <= 5e-3, (key, np.median(rel), rel.max())
        # the first pass is the same code in both builds: its trace row (J_hat, g_norm, step size, J, residual) is the same bits -- except
        # for an instance that ONE of the two hands to the wave kernel (a line search that rejects every step at the noise floor of the
        # unreachable atol): that kernel re-solves it from the start and rewrites its rows
        first_same = (re["trace"][:, 0].nan_to_num(-7.0) == full["trace"][:, 0].nan_to_num(-7.0)).all(dim=1)
        assert float(first_same.float().mean()) >= (1.0 if not kwargs else 0.9)
        # later passes: g_norm (computed from the vector recursion's k_t) agrees to rounding relative to the first pass's scale
        from tfmpc.solvers.ilqr import TRACE_COLUMNS
        gcol = TRACE_COLUMNS.index("g_norm")
        g0 = full["trace"][:, 0, gcol]
        for r in range(1, min_passes):
            ok = same & (re["trace_len"] > r) & (full["trace_len"] > r)
            dg = (re["trace"][:, r, gcol] - full["trace"][:, r, gcol]).abs()[ok]
            assert float((dg / g0[ok]).max()) <= 1e-4, (r, float((dg / g0[ok]).max()))
    # and against the fp64 restatement of ilqr.py, like every other kernel
    solver = iLQR(LQEnv(F, f, C, c))
    res = solver.solve_device(x0[..., None], T, u_init=u0)
    for b in (1, B - 2):
        o = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b]))
        x, u, cs, it = o.solve(x0[b], T, u_init=u0[b])
        assert it == int(res["iterations"][b])
        o32 = ilqr_ref.ILQRRef(envs_ref.LQEnv(F[b], f[b], C[b], c[b], dtype=np.float32), dtype=np.float32)
        x32, _, _, _ = o32.solve(x0[b], T, u_init=u0[b])
        budget = 5 * max(np.abs(x32 - x).max(), 1e-5 * np.abs(x).max())
        assert np.abs(res["states"][b, ..., 0].cpu().numpy() - x).max() <= budget
