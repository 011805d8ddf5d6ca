"""Argument validation of every C-ABI entry point (include/tfmpc_hip.h).  Bad sizes and null
pointers must come back as TFMPC_ERR_* BEFORE anything touches the GPU, so these calls are safe
(and meaningful) on a machine without one."""

import ctypes

import pytest

from tfmpc import _hip

ERR_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE = -1, -2, -4
NULL = None


@pytest.fixture(scope="module")
def lib():
    return _hip.load()


def _dummy():
    """A non-null pointer that is never dereferenced (validation fails first)."""
    return ctypes.c_void_p(0x1000)


def test_lqr_entry_points_reject_bad_arguments(lib):
    d = _dummy()
    ops = [d, 0, d, 0, d, 0, d, 0]
    assert lib.tfmpc_lqr_solve_f32(-1, 3, 2, 5, *ops, d, d, d, d, d, d, d, d, d, d, d, 0, NULL) == ERR_ARG
    assert lib.tfmpc_lqr_solve_f32(4, 0, 2, 5, *ops, d, d, d, d, d, d, d, d, d, d, d, 0, NULL) == ERR_ARG
    assert lib.tfmpc_lqr_solve_f32(4, 3, 2, 5, NULL, 0, d, 0, d, 0, d, 0, d, d, d, d, d, d, d, d, d, d, d, 0, NULL) == ERR_ARG
    assert lib.tfmpc_lqr_solve_f32(4, 3, 2, 5, *ops, NULL, d, d, d, d, d, d, d, d, d, d, 0, NULL) == ERR_ARG      # x0
    # gains not requested and no workspace -> workspace error, not a crash
    assert lib.tfmpc_lqr_solve_f32(4, 3, 2, 5, *ops, d, d, d, d, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) == ERR_WORKSPACE
    assert lib.tfmpc_lqr_solve_f32(4, 200, 200, 5, *ops, d, d, d, d, d, d, d, d, d, d, d, 0, NULL) == ERR_UNSUPPORTED
    assert lib.tfmpc_lqr_backward_f32(4, 3, 2, 5, *ops, NULL, d, d, d, d, d, NULL) == ERR_ARG                     # K required
    assert lib.tfmpc_lqr_forward_f32(4, 3, 2, 5, *ops, d, 0, d, 0, NULL, d, d, d, NULL) == ERR_ARG                # x0
    assert lib.tfmpc_lqr_forward_f32(0, 3, 2, 5, *ops, d, 0, d, 0, d, d, d, d, NULL) == 0                         # empty batch is fine
    assert lib.tfmpc_lqr_workspace_bytes(0, 3, 2, 5) == 0


def _env(kind=_hip.ENV_NAVLQR, n=2, m=2, with_params=True):
    env = _hip.TfmpcEnv()
    env.kind, env.n, env.m = kind, n, m
    env.low = env.high = 0x1000
    if with_params:
        for i in range(_hip.ENV_MAX_PARAMS):
            env.p[i] = 0x1000
    return env


def test_ilqr_entry_points_reject_bad_arguments(lib):
    d = _dummy()
    cfg = _hip.TfmpcIlqrConfig()
    cfg.max_iterations, cfg.n_alphas = 10, 11
    good = _env()
    byref = ctypes.byref
    assert lib.tfmpc_ilqr_rollout_f32(NULL, 4, 5, d, d, d, d, NULL) == ERR_ARG
    assert lib.tfmpc_ilqr_rollout_f32(byref(_env(kind=9)), 4, 5, d, d, d, d, NULL) == ERR_ARG
    assert lib.tfmpc_ilqr_rollout_f32(byref(_env(n=2, m=3)), 4, 5, d, d, d, d, NULL) == ERR_ARG     # reference envs: m == n
    assert lib.tfmpc_ilqr_rollout_f32(byref(_env(with_params=False)), 4, 5, d, d, d, d, NULL) == ERR_ARG
    assert lib.tfmpc_ilqr_rollout_f32(byref(good), 4, 5, NULL, d, d, d, NULL) == ERR_ARG
    assert lib.tfmpc_ilqr_rollout_f32(byref(good), -4, 5, d, d, d, d, NULL) == ERR_ARG
    assert lib.tfmpc_ilqr_rollout_f32(byref(_env(kind=_hip.ENV_HVAC, n=300, m=300)), 4, 5, d, d, d, d, NULL) == ERR_UNSUPPORTED
    assert lib.tfmpc_ilqr_derivatives_f32(byref(good), 4, 5, NULL, d, *([d] * 13), NULL) == ERR_ARG
    assert lib.tfmpc_ilqr_forward_f32(byref(good), 4, 5, d, d, d, d, NULL, 0, d, d, d, d, d, NULL) == ERR_ARG   # alpha
    bw = [d] * 12
    assert lib.tfmpc_ilqr_backward_f32(4, 2, 2, 5, *bw, d, d, 0, NULL, 0, d, d, d, d, d, d, NULL) == ERR_ARG      # mu
    assert lib.tfmpc_ilqr_backward_f32(4, 0, 2, 5, *bw, d, d, 0, d, 0, d, d, d, d, d, d, NULL) == ERR_ARG
    bad_cfg = _hip.TfmpcIlqrConfig()
    bad_cfg.max_iterations, bad_cfg.n_alphas = 10, 99
    assert lib.tfmpc_ilqr_solve_f32(byref(good), byref(bad_cfg), 4, 5, d, d, d, d, d, d, d, d, 1 << 20, NULL) == ERR_ARG
    assert lib.tfmpc_ilqr_solve_f32(byref(good), NULL, 4, 5, d, d, d, d, d, d, d, d, 1 << 20, NULL) == ERR_ARG
    assert lib.tfmpc_ilqr_solve_f32(byref(good), byref(cfg), 4, 5, d, d, d, d, d, d, d, d, 16, NULL) == ERR_WORKSPACE
    assert lib.tfmpc_ilqr_solve_f32(byref(good), byref(cfg), 4, 5, d, d, d, d, d, NULL, d, d, 1 << 20, NULL) == ERR_ARG   # iterations
    # five slabs per instance (K, k, candidate x, u, costs), rounded up to 256 bytes ...
    slabs = lambda B, n, m, T: -(-B * (T * m * n + T * m + (T + 1) * n + T * m + (T + 1)) * 4 // 256) * 256
    # ... plus, for the shapes the matrix-core LQ kernel takes (n <= 16, m <= 8, n + m > 6), -Q_uu(t)^-1 of its first backward pass: 64 floats
    # per instance and time step, which the later, gain-reusing passes read (round 6)
    assert lib.tfmpc_ilqr_workspace_bytes(4, 3, 5, 5) == slabs(4, 3, 5, 5) + 4 * 5 * 64 * 4
    # ... plus, for n == m <= 32, the wave-major trajectory buffers of the 16-per-wave costate kernel: per wave two
    # buffers of (T + 1) + T tiles of 64 lanes x 4 rows and (T + 1) x 64 stage costs, T x 64 selector bytes, one time step's
    # worth of "nowhere" for the counted unconditional stores, 8 x 7 checkpoint tile sets of the multi-wave groups; and, with one tile, behind
    # the slices of all groups, three 16-byte pieces per lane and time step for the coefficients of the two-part costate sweep -- for at most 512
    # groups, the launches whose forms read them (round 5; round 6: no longer in every group's slice)
    wave = lambda tiles, T: (2 * ((T + 1) * tiles * 256 + T * tiles * 256 + (T + 1) * 64) * 4 + T * 64 + 255
                             + (tiles * 256 + 64) * 4 + 8 * 7 * tiles * 256 * 4) // 256 * 256
    coef = lambda groups, T: min(groups, 512) * (T * 3 * 256 * 4)
    assert lib.tfmpc_ilqr_workspace_bytes(4, 3, 3, 5) == slabs(4, 3, 3, 5) + 1 * wave(1, 5) + coef(1, 5)      # n <= 4: 64 instances per wave
    assert lib.tfmpc_ilqr_workspace_bytes(40, 32, 32, 7) == slabs(40, 32, 32, 7) + 3 * wave(2, 7)  # 16 per wave, two tiles
    big = lib.tfmpc_ilqr_workspace_bytes(65536, 4, 4, 100)                                          # 1 024 groups: the slab is capped
    assert big == slabs(65536, 4, 4, 100) + 65536 * 100 * 64 * 4 + 1024 * wave(1, 100) + coef(1024, 100)      # (n + m > 6: the LQ kernel's slab too)
    # ... plus, for the 2-D envs, one scratch block of line-search candidates per wavefront (5 T + 1 rows of 64 lanes)
    assert lib.tfmpc_ilqr_workspace_bytes(4, 2, 2, 5) >= slabs(4, 2, 2, 5) + 4 * (5 * 5 + 1) * 64 * 4
    assert lib.tfmpc_ilqr_workspace_bytes(0, 2, 2, 5) == 0
    # the size for ONE env (round 6): the slabs + only what that env kind's kernels read -- and the solve accepts it
    for kind, extra in ((_hip.ENV_LQ, 65536 * 100 * 64 * 4), (_hip.ENV_RESERVOIR, 1024 * wave(1, 100) + coef(1024, 100)), (_hip.ENV_HVAC, 1024 * wave(1, 100) + coef(1024, 100))):
        e = _env(kind=kind, n=4, m=4)
        assert lib.tfmpc_ilqr_workspace_bytes_for(byref(e), 65536, 100) == slabs(65536, 4, 4, 100) + extra, kind
    nav = lib.tfmpc_ilqr_workspace_bytes_for(byref(_env(kind=_hip.ENV_NAVIGATION, n=2, m=2)), 4, 5)
    assert slabs(4, 2, 2, 5) + 4 * (5 * 5 + 1) * 64 * 4 <= nav < lib.tfmpc_ilqr_workspace_bytes(4, 2, 2, 5)       # (its scratch, not the costate kernel's buffers)
    assert lib.tfmpc_ilqr_workspace_bytes_for(NULL, 4, 5) == 0
    assert lib.tfmpc_boxqp_f32(4, 0, d, d, d, d, d, d, d, d, NULL) == ERR_ARG
    assert lib.tfmpc_boxqp_f32(4, 3, NULL, d, d, d, d, d, d, d, NULL) == ERR_ARG
    assert lib.tfmpc_boxqp_f32(0, 3, d, d, d, d, d, d, d, d, NULL) == 0


def test_every_documented_option_is_known_to_the_library():
    """tfmpc_set_option / tfmpc_get_option (host code only): every name include/tfmpc_hip.h and INTEGRATION.md document is accepted, the value read
    back is the value set, an unknown name and an over-long value are argument errors, and the previous override is restored."""
    names = ("TFMPC_LQR_KERNEL", "TFMPC_LQR_MFMA", "TFMPC_ILQR_KERNEL", "TFMPC_COSTATE_WAVES", "TFMPC_ILQR_RETRY", "TFMPC_COSTATE_COUPLING",
             "TFMPC_LQR_WAVES", "TFMPC_BOX_HELPERS", "TFMPC_BOX_HELP_AFTER", "TFMPC_ILQR_LQ_REUSE", "TFMPC_GROUP_STORED", "TFMPC_BOX_SPECULATE")
    import os
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "tfmpc_hip.h")).read()
    for name in names:
        assert name in header, name                                    # documented where the ABI is
        before = _hip.get_option(name)
        with _hip.option(name, "7"):
            assert _hip.get_option(name) == "7"
        assert _hip.get_option(name) == before
    with pytest.raises(ValueError):
        _hip.set_option("TFMPC_NO_SUCH_OPTION", "1")
    assert _hip.load().tfmpc_set_option(b"TFMPC_BOX_HELPERS", b"x" * 40) == ERR_ARG
