"""Lane-per-instance LQR kernel (tiny shapes, n + m <= 6): parity with the fp64 oracle and
agreement with the wave-per-instance kernel on the same inputs."""

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


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (2, 2), (3, 2), (3, 3), (4, 2)])
@pytest.mark.parametrize("T", [1, 10])
def test_lane_kernel_matches_oracle_and_wave_kernel(force_kernel, n, m, T):
    B = 200                                   # not a multiple of 64: exercises the tail wave
    F, f, C, c, x0 = problems.make_lqr_batch(B, n, m, seed0=7000 + 10 * n + m)
    ref64 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float64, want_policy=True, want_value=True)
    ref32 = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float32, want_policy=True, want_value=True)
    lqr = LQR(F, f, C, c)
    outs = {}
    for kern in ("lane", "generic"):
        force_kernel(kern)
        outs[kern] = lqr.solve_device(x0, T, want_policy=True, want_value=True)
        torch.cuda.synchronize()
        assert int(outs[kern]["status"].abs().sum()) == 0
    for key in ("states", "actions", "costs", "K", "k", "V", "v", "const"):
        got = outs["lane"][key].cpu().numpy().astype(np.float64).reshape(ref64[key].shape)
        wave = outs["generic"][key].cpu().numpy().astype(np.float64).reshape(ref64[key].shape)
        ratios = []
        for b in range(B):
            scale = np.abs(ref64[key][b]).max()
            e32 = max(np.abs(ref32[key][b].astype(np.float64) - ref64[key][b]).max(), 1e-6 * scale)
            ratios.append(np.abs(got[b] - ref64[key][b]).max() / e32)
        assert np.median(ratios) <= 2.0 and np.quantile(ratios, 0.9) <= BUDGET and max(ratios) <= 5 * BUDGET, (key, max(ratios))
        # same reference op order in both kernels: they agree to fp32 rounding of the largest entry
        assert np.abs(got - wave).max() <= 1e-3 * max(np.abs(wave).max(), 1.0), key


def test_lane_backward_then_forward_equals_fused(force_kernel):
    force_kernel("lane")
    B, n, m, T = 100, 2, 2, 12
    F, f, C, c, x0 = problems.make_lqr_batch(B, n, m, seed0=9100)
    lqr = LQR(F, f, C, c)
    fused = lqr.solve_device(x0, T, want_policy=True, want_value=True)
    policy, value_fn = lqr.backward(T)
    xs, us, cs = lqr.forward(policy, x0[..., None], T)
    assert torch.equal(policy.K, fused["K"]) and torch.equal(value_fn.const, fused["const"])
    assert torch.equal(xs, fused["states"]) and torch.equal(cs, fused["costs"])
