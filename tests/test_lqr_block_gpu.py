"""Workgroup-per-instance LQR kernel for large shapes (lqr_block.hip, ``-m gpu``): parity with the fp64 C
restatement within the same budget as the other LQR kernels (error relative to the fp32 restatement's own
error, tests/test_lqr_gpu.py), agreement with the wave-per-instance kernel, value-function outputs, the split
backward / forward entry points, shared operands."""

import os

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


@pytest.mark.parametrize("n,m,T", [(32, 16, 12), (24, 24, 8), (17, 9, 15), (33, 3, 6), (20, 1, 9), (5, 19, 7), (48, 16, 4),
                                   (3, 2, 10), (16, 16, 1), (20, 4, 130)])       # T = 130: three chunks of the cost post-pass
def test_block_kernel_matches_oracle_and_wave_kernel(force_kernel, n, m, T):
    B = 37
    F, f, C, c, x0 = _problem(B, n, m, seed=97 * n + m)
    ref64 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float64, want_policy=True, want_value=True)
    ref32 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float32, want_policy=True, want_value=True)
    lqr = LQR(F, f, C, c)
    outs = {}
    for kern in ("block", "generic"):
        force_kernel(kern)
        outs[kern] = lqr.solve_device(x0, T, want_policy=True, want_value=True)
        torch.cuda.synchronize()
        assert int(outs[kern]["status"].abs().sum()) == 0
    for key in ("states", "actions", "costs", "K", "k", "V", "v", "const"):
        got = outs["block"][key].cpu().numpy().astype(np.float64).reshape(ref64[key].shape)
        wave = outs["generic"][key].cpu().numpy().astype(np.float64).reshape(ref64[key].shape)
        assert np.isfinite(got).all()
        ratios = []
        for b in range(B):
            scale = np.abs(ref64[key][b]).max()
            e32 = max(np.abs(ref32[key][b].astype(np.float64) - ref64[key][b]).max(), 1e-6 * scale)
            ratios.append(np.abs(got[b] - ref64[key][b]).max() / e32)
        assert np.median(ratios) <= 2.0 and np.quantile(ratios, 0.9) <= BUDGET and max(ratios) <= 5 * BUDGET, (key, max(ratios))
        assert np.abs(got - wave).max() <= 1e-3 * max(np.abs(wave).max(), 1.0), key


def test_block_kernel_is_the_default_for_large_shapes_and_split_equals_fused(force_kernel):
    force_kernel(None)
    lib = _hip.require_gpu()
    assert lib.tfmpc_lqr_kernel_name(40, 16, 10) == b"block_mfma_f32"
    assert lib.tfmpc_lqr_kernel_name(32, 16, 10) == b"mfma_32x16"            # round 2: 2 x 2 tiles on the bf16 matrix cores
    assert lib.tfmpc_lqr_kernel_name(16, 8, 10).startswith(b"mfma_16x8")
    assert lib.tfmpc_lqr_kernel_name(5, 19, 10) == b"block_mfma_f32 (batch <= 2048) / generic_wave"
    assert lib.tfmpc_lqr_kernel_name(17, 2, 10) == b"mfma_32x16 (zero-padded)"
    B, n, m, T = 21, 36, 12, 9
    F, f, C, c, x0 = _problem(B, n, m, seed=5)
    lqr = LQR(F, f, C, c)
    fused = lqr.solve_device(x0, T, want_policy=True, want_value=True)
    policy, value_fn = lqr.backward(T)
    xs, us, cs = lqr.forward(policy, x0[..., None], T)
    assert torch.equal(policy.K, fused["K"]) and torch.equal(policy.k, fused["k"])
    assert torch.equal(value_fn.V, fused["V"]) and torch.equal(value_fn.const, fused["const"])
    assert torch.equal(xs, fused["states"]) and torch.equal(us, fused["actions"]) and torch.equal(cs, fused["costs"])


def test_block_kernel_with_shared_dynamics_and_costs(force_kernel):
    force_kernel("block")
    B, n, m, T = 19, 24, 8, 6
    F, f, C, c, x0 = _problem(B, n, m, seed=11)
    shared = LQR(F[0], f[0], C[0], c[0]).solve_device(x0, T)            # one model, B initial states
    tiled = LQR(np.repeat(F[:1], B, 0), np.repeat(f[:1], B, 0), np.repeat(C[:1], B, 0), np.repeat(c[:1], B, 0)).solve_device(x0, T)
    for key in ("states", "actions", "costs"):
        assert torch.equal(shared[key], tiled[key]), key
