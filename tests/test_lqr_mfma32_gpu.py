"""2 x 2-tile matrix-core LQR kernel for shapes beyond 16 x 8 up to n = 32, m = 16 (tf-mpc_amd/csrc/lqr_mfma32x16.hip, ``-m gpu``):
parity with the fp64 C restatement of /root/reference/tfmpc/solvers/lqr.py:59-166 within the budget of the other LQR kernels
(error relative to the fp32 restatement's own error), value-function outputs, agreement with the wave kernel, the split
backward / forward entry points, padded shapes, and the reference's make_lqr spectrum at n = 32, m = 16."""

import numpy as np
import pytest
import torch

import problems
from oracle import c_oracle
from tfmpc import _hip
from tfmpc.solvers.lqr import LQR

pytestmark = pytest.mark.gpu
BUDGET = 5.0


@pytest.fixture
def force_kernel():
    def set_(name):
        _hip.set_option("TFMPC_LQR_KERNEL", name)
    yield set_
    set_(None)


def _problem(B, n, m, seed):
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=seed)
    F *= 1.5 / np.sqrt(n)            # spectral radius ~1.5: unstable open loop, well inside fp32 for the Riccati sweep
    return F, f, C, c, x0


@pytest.mark.parametrize("n,m,T", [(32, 16, 12), (32, 16, 50), (24, 12, 8), (17, 9, 15), (32, 3, 6), (20, 16, 9), (9, 16, 7),
                                   (28, 12, 60), (16, 9, 5), (3, 11, 4)])
def test_mfma32_kernel_matches_oracle_and_wave_kernel(force_kernel, n, m, T):
    lib = _hip.require_gpu()
    assert lib.tfmpc_lqr_kernel_name(n, m, T).startswith(b"mfma_32x16")
    B = 37
    F, f, C, c, x0 = _problem(B, n, m, seed=97 * n + m)
    ref64 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float64, want_policy=True, want_value=True)
    ref32 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float32, want_policy=True, want_value=True)
    lqr = LQR(F, f, C, c)
    outs = {}
    for kern in (None, "generic"):
        force_kernel(kern)
        outs[kern] = lqr.solve_device(x0, T, want_policy=True, want_value=True)
        torch.cuda.synchronize()
        assert int(outs[kern]["status"].abs().sum()) == 0, kern
    for key in ("states", "actions", "costs", "K", "k", "V", "v", "const"):
        got = outs[None][key].cpu().numpy().astype(np.float64).reshape(ref64[key].shape)
        wave = outs["generic"][key].cpu().numpy().astype(np.float64).reshape(ref64[key].shape)
        assert np.isfinite(got).all(), key
        ratios = []
        for b in range(B):
            scale = np.abs(ref64[key][b]).max()
            e32 = max(np.abs(ref32[key][b].astype(np.float64) - ref64[key][b]).max(), 1e-6 * scale)
            ratios.append(np.abs(got[b] - ref64[key][b]).max() / e32)
        assert np.median(ratios) <= 2.0 and np.quantile(ratios, 0.9) <= BUDGET and max(ratios) <= 5 * BUDGET, (key, np.median(ratios), max(ratios))
        assert np.abs(got - wave).max() <= 1e-3 * max(np.abs(wave).max(), 1.0), key
    # no value function requested: the same trajectories, bit for bit (the gains then live in the workspace)
    force_kernel(None)
    plain = lqr.solve_device(x0, T)
    assert torch.equal(plain["states"], outs[None]["states"]) and torch.equal(plain["costs"], outs[None]["costs"])
    # split entry points == fused
    policy, value_fn = lqr.backward(T)
    xs, us, cs = lqr.forward(policy, x0[..., None], T)
    assert torch.equal(policy.K, outs[None]["K"]) and torch.equal(policy.k, outs[None]["k"])
    assert torch.equal(value_fn.V, outs[None]["V"]) and torch.equal(value_fn.const, outs[None]["const"])
    assert torch.equal(xs, outs[None]["states"]) and torch.equal(us, outs[None]["actions"]) and torch.equal(cs, outs[None]["costs"])


def test_mfma32_on_the_reference_spectrum_at_n32_m16():
    """make_lqr(32, 16): C = make_spd_matrix(48) (eigenvalues ~1e-3 .. 48), F ~ N(0, 1) -- the reference's generator at
    BASELINE configs[4]'s literal dims; all eight outputs, error distribution against the fp32 restatement's."""
    B, n, m, T = 256, 32, 16, 50
    F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=77)
    ref64 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float64, nthreads=8, want_policy=True, want_value=True)
    ref32 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float32, nthreads=8, want_policy=True, want_value=True)
    out = LQR(F, f, C, c).solve_device(x0, T, want_policy=True, want_value=True)
    torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    for key in ("states", "actions", "costs", "K", "k", "V", "v", "const"):
        got = out[key].cpu().numpy().astype(np.float64).reshape(ref64[key].shape)
        scale = np.abs(ref64[key]).reshape(B, -1).max(axis=1)
        e_dev = np.abs(got - ref64[key]).reshape(B, -1).max(axis=1) / scale
        e_32 = np.maximum(np.abs(ref32[key].astype(np.float64) - ref64[key]).reshape(B, -1).max(axis=1) / scale, 1e-7)
        assert np.median(e_dev) <= 1.5 * np.median(e_32), (key, np.median(e_dev), np.median(e_32))
        assert np.quantile(e_dev / e_32, 0.9) <= 5.0 and (e_dev / e_32).max() <= 25.0, (key, np.quantile(e_dev / e_32, 0.9), (e_dev / e_32).max())


def test_mfma32_flags_a_non_pd_quu():
    n, m = 20, 10
    F = np.zeros((n, n + m), dtype=np.float32)
    C = np.zeros((n + m, n + m), dtype=np.float32)     # Q_uu == 0 -> singular
    out = LQR(F, np.zeros(n), C, np.zeros(n + m)).solve_device(np.ones(n), 3)
    torch.cuda.synchronize()
    assert int(out["status"][0]) & (_hip.ST_SINGULAR | _hip.ST_NOT_PD)
