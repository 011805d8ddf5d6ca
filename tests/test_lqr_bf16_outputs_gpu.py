"""Policy / value-function outputs of the LQR solve in 16-bit containers (what they hold:
/root/reference/tfmpc/solvers/lqr.py:107-129): ``LQR.solve_device(..., storage_bf16=True)`` = ``tfmpc_lqr_solve_bf16out_f32``.
Contract (include/tfmpc_hip.h): every 16-bit value is the fp32 output rounded to nearest even -- checked BIT FOR BIT
against the rounded outputs of the fp32 entry point -- and the trajectory is the fp32 one, bit for bit (the rollout reads
the fp32 gains).  Shapes: the headline kernel (16 x 8, exact and zero-padded), the 2 x 2-tile kernel (32 x 16 and padded),
the wave kernel (the lane / workgroup kernels' shapes are routed to it).  Then the error of the 16-bit value function
against the fp64 oracle: the bf16 rounding (2^-9 relative) on top of the fp32 result, nothing else."""

import numpy as np
import pytest
import torch

import problems
from oracle import c_oracle
from tfmpc import _hip
from tfmpc.solvers.lqr import LQR

pytestmark = pytest.mark.gpu


def _rne_bf16(t):
    bits = t.contiguous().view(torch.int32)
    return ((bits + 0x7FFF + ((bits >> 16) & 1)) >> 16).to(torch.int16)


@pytest.mark.parametrize("n,m,T,B", [(16, 8, 50, 300), (13, 5, 20, 70), (32, 16, 30, 40), (24, 12, 12, 33), (3, 2, 10, 64), (40, 20, 6, 5)])
def test_sixteen_bit_outputs_are_the_rounded_fp32_outputs(n, m, T, B):
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=n + m)
    F = F * 0.5 * np.sqrt(16.0 / n)
    lqr = LQR(F, f[..., None], C, c[..., None])
    x0 = x0.astype(np.float32)[..., None]
    # shapes of the lane / workgroup kernels take the wave kernel for 16-bit outputs: the fp32 reference call is made on
    # that kernel too (different kernels agree to rounding, not bit for bit)
    wave_shape = n + m <= 6 or n > 32 or m > 16
    with _hip.option("TFMPC_LQR_KERNEL", "generic" if wave_shape else _hip.get_option("TFMPC_LQR_KERNEL")):
        ref = lqr.solve_device(x0, T, want_policy=True, want_value=True)
    out = lqr.solve_device(x0, T, want_policy=True, want_value=True, storage_bf16=True)
    torch.cuda.synchronize()
    assert int(out["status"].abs().sum()) == 0
    for key in ("states", "actions", "costs"):
        assert torch.equal(out[key], ref[key]), key                   # the trajectory is the fp32 one
    for key in ("K", "k", "V", "v", "const"):
        assert out[key].dtype == torch.bfloat16 and out[key].shape == ref[key].shape
        assert torch.equal(out[key].view(torch.int16), _rne_bf16(ref[key])), key
    # a subset of the outputs, and no fp32 policy at all: same values
    only_v = lqr.solve_device(x0, T, want_value=True, storage_bf16=True)
    torch.cuda.synchronize()
    assert torch.equal(only_v["V"].view(torch.int16), out["V"].view(torch.int16)) and "K" not in only_v
    assert torch.equal(only_v["states"], ref["states"])


def test_value_function_error_of_the_sixteen_bit_containers_against_fp64():
    """n = 16, m = 8, T = 50 on the reference's spectrum (make_lqr): relative to the tensor's max-abs, the 16-bit value
    function is the bf16 rounding step (<= 2^-9 of the largest entry) away from the fp64 oracle; the fp32 one ~1e-4."""
    n, m, T, B = 16, 8, 50, 256
    F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=5)
    lqr = LQR(F, f[..., None], C, c[..., None])
    out = lqr.solve_device(x0.astype(np.float32)[..., None], T, want_policy=True, want_value=True, storage_bf16=True)
    torch.cuda.synchronize()
    ref = c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float64, nthreads=8, want_policy=True, want_value=True)
    for key, name in (("K", "K"), ("k", "k"), ("V", "V"), ("v", "v"), ("const", "const")):
        got = out[key].float().cpu().numpy().reshape(ref[name].shape)
        scale = np.abs(ref[name]).reshape(B, -1).max(axis=1)
        err = np.abs(got - ref[name]).reshape(B, -1).max(axis=1) / scale
        assert np.median(err) <= 2.0 ** -8 and err.max() <= 2.0 ** -7, (key, np.median(err), err.max())
