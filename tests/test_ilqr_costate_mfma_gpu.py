"""Sixteen instances per wavefront with the coupling-matrix products on the matrix cores
(tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip; HVAC / Reservoir envs shared by the batch, `TFMPC_ILQR_KERNEL=costate_mfma`,
the default for large batches) against the generic wave-per-instance kernel and the fp64 oracle.

Reservoir keeps the wave kernels' operation order and reduction trees, and on a 0/1 `downstream` matrix (every
reference config) a row of the coupling product is a single exact term: there every output must be BIT-identical, which
is the only meaningful comparison on this env (any rounding difference flips line-search decisions and changes the
trajectories completely).  HVAC folds the linear part of the room balance into the matrix, so it agrees to fp32
rounding until a bang-bang decision flips: most instances to ~1e-7, a few per cent differently but equally good."""

import os

import numpy as np
import pytest
import torch

import problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_kernel():
    def set_(name):
        _hip.set_option("TFMPC_ILQR_KERNEL", name)
    yield set_
    set_(None)


def _env(kind, n, B, seed):
    rng = np.random.default_rng(100 + n)
    if kind == "hvac":
        return HVAC.load(dict(problems.hvac_config(n, seed=seed))), rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32)
    return Reservoir.load(dict(problems.reservoir_config(n, seed=seed))), rng.uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)


def _both(force_kernel, solver, x0, T, u0, kernels=("wave", "costate_mfma")):
    out = {}
    for kern in kernels:
        force_kernel(kern)
        out[kern] = {k: v.clone() for k, v in solver.solve_device(x0, T, u_init=u0).items() if torch.is_tensor(v) and k != "workspace"}
        torch.cuda.synchronize()
    return out


SHAPES = [(32, 24, 70), (21, 13, 9), (17, 7, 5), (30, 40, 33), (32, 1, 1), (18, 2, 3), (16, 9, 7), (12, 11, 4),
          (6, 20, 40), (4, 15, 33), (2, 5, 3), (3, 1, 2), (8, 12, 130), (5, 30, 257), (7, 3, 1), (28, 30, 16)]


@pytest.mark.parametrize("n,T,B", SHAPES)
def test_reservoir_is_bit_identical_to_the_wave_kernel(force_kernel, n, T, B):
    env, x0 = _env("reservoir", n, B, n)
    solver = iLQR(env, max_iterations=6)
    u0 = solver.random_actions(T, B, seed=n)
    out = _both(force_kernel, solver, x0, T, u0)
    for key in ("iterations", "status", "states", "actions", "costs"):
        assert torch.equal(out["costate_mfma"][key], out["wave"][key]), key
    assert bool(torch.isfinite(out["costate_mfma"]["costs"]).all())


@pytest.mark.parametrize("n,T,B,iters,atol", [(32, 100, 512, 12, None), (8, 25, 203, 30, 0.05), (20, 25, 100, 30, 0.05)])
def test_reservoir_columns_finish_at_different_times(force_kernel, n, T, B, iters, atol):
    """The 16 instances of a wave run in lockstep; columns that converge early (loose atol) keep executing with masked
    stores, and the nominal / candidate buffers flip per column.  Still bit-identical to the wave kernel."""
    env, x0 = _env("reservoir", n, B, n)
    kw = {} if atol is None else {"atol": atol}
    solver = iLQR(env, max_iterations=iters, **kw)
    u0 = solver.random_actions(T, B, seed=n)
    out = _both(force_kernel, solver, x0, T, u0)
    if atol is not None:
        assert len(torch.unique(out["wave"]["iterations"])) >= 4       # the scenario does spread the finishing times
    for key in ("iterations", "status", "states", "actions", "costs"):
        assert torch.equal(out["costate_mfma"][key], out["wave"][key]), key


@pytest.mark.parametrize("n,B", [(24, 4200), (10, 4100), (24, 40), (10, 40), (4, 3)])
def test_default_dispatch(force_kernel, n, B):
    """A shared env goes to the 16-per-wave kernel without any forcing at n <= 16 (any batch size), on Reservoir at any n, and
    on HVAC at n > 16 from 4097 instances (smaller large-n HVAC batches stay on the register-resident kernels): on Reservoir
    every kernel equals the wave kernel bit for bit, whichever is dispatched."""
    env, x0 = _env("reservoir", n, B, 3)
    solver = iLQR(env, max_iterations=4)
    u0 = solver.random_actions(10, B, seed=2)
    out = _both(force_kernel, solver, x0, 10, u0, kernels=("wave", None, "costate_mfma"))
    for key in ("iterations", "states", "actions", "costs"):
        assert torch.equal(out[None][key], out["wave"][key]) and torch.equal(out["costate_mfma"][key], out["wave"][key])


@pytest.mark.parametrize("n,T,B", SHAPES)
def test_hvac_tracks_the_wave_kernel(force_kernel, n, T, B):
    env, x0 = _env("hvac", n, B, n)
    solver = iLQR(env, max_iterations=6)
    u0 = solver.random_actions(T, B, seed=n)
    out = _both(force_kernel, solver, x0, T, u0)
    w, f = out["wave"], out["costate_mfma"]
    assert bool(torch.isfinite(f["costs"]).all())
    assert torch.equal(w["status"], f["status"])
    rel = ((w["states"] - f["states"]).abs().flatten(1).max(dim=1).values /
           w["states"].abs().flatten(1).max(dim=1).values)
    close = rel < 2e-6                                            # fp32 rounding through <= 6 iterations
    # the rest: a bang-bang decision flipped (at most a tenth of the batch; one instance of a tiny batch)
    assert int((~close).sum()) <= max(1, B // 10), float(rel.max())
    assert bool((w["iterations"] == f["iterations"])[close].all())
    tw, tf_ = w["costs"].sum(dim=1), f["costs"].sum(dim=1)
    assert float(((tw - tf_).abs() / tw.abs()).max()) < 1e-3       # flipped or not, the solutions are equally good


def test_hvac_first_iterations_match_the_fp64_oracle_at_n24(force_kernel):
    """The oracle test of the register-resident kernel (tests/test_ilqr_adjoint_gpu.py), for the 16-per-wave kernel:
    16 different start states in one wave, each against the fp64 restatement of ilqr.py + hvac/__init__.py."""
    from oracle import envs_ref, ilqr_ref
    n, T, B = 24, 12, 16
    cfg = problems.hvac_config(n, seed=7)
    rng = np.random.default_rng(1)
    x0 = rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32)
    solver = iLQR(HVAC.load(dict(cfg)), max_iterations=2)
    u0 = solver.random_actions(T, B, seed=3)
    force_kernel("costate_mfma")
    out = solver.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    states, its = out["states"].cpu().numpy()[..., 0], out["iterations"].cpu().numpy()
    o = ilqr_ref.ILQRRef(envs_ref.HVAC(**cfg, dtype=np.float64), dtype=np.float64, max_iterations=2)
    for b in (0, 5, 15):
        xs, us, cs, it64 = o.solve(x0[b].astype(np.float64), T, u_init=u0[b].cpu().numpy().astype(np.float64))
        assert int(its[b]) == it64
        assert np.abs(states[b] - np.asarray(xs).reshape(T + 1, n)).max() <= 1e-4 * np.abs(xs).max()


@pytest.mark.parametrize("kind", ["hvac", "reservoir"])
def test_dense_couplings_solve_as_well_as_the_wave_kernel(force_kernel, kind):
    """Dense random couplings: the matrix-core row sums round differently from the wave kernel's, so trajectories may
    part ways at a flipped decision -- but the first rollout (no decisions yet) agrees to rounding and the solves end
    at costs of the same quality."""
    n, T, B = 27, 12, 64
    rng = np.random.default_rng(50 + n)
    if kind == "reservoir":
        cfg = dict(problems.reservoir_config(n, seed=n))
        # every reservoir spills into every other one, about half of its outflow in all (a stable network)
        cfg["downstream"] = (rng.uniform(0.0, 1.0, size=(n, n)) / n).astype(np.float32) * (1.0 - np.eye(n, dtype=np.float32))
        env, x0 = Reservoir.load(cfg), rng.uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)
    else:
        cfg = dict(problems.hvac_config(n, seed=n))
        cfg["adj"] = np.triu(np.ones((n, n), dtype=bool), 1)
        env, x0 = HVAC.load(cfg), rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32)
    u0 = iLQR(env).random_actions(T, B, seed=n)
    start = _both(force_kernel, iLQR(env, max_iterations=1, atol=1e9), x0, T, u0)     # "converged" at once: the start rollout
    rel = (start["wave"]["states"] - start["costate_mfma"]["states"]).abs().max() / start["wave"]["states"].abs().max()
    assert float(rel) < 1e-5
    full = _both(force_kernel, iLQR(env, max_iterations=5), x0, T, u0)
    tw, tf_ = full["wave"]["costs"].sum(dim=1), full["costate_mfma"]["costs"].sum(dim=1)
    assert bool(torch.isfinite(tf_).all())
    assert float(tf_.median()) <= float(tw.median()) * 1.05 + 1e-3                # costs are positive on both envs


@pytest.mark.parametrize("kind,n,B", [("hvac", 32, 300), ("reservoir", 32, 300), ("hvac", 6, 100), ("reservoir", 4, 100), ("hvac", 7, 33)])
def test_sixteen_bit_trajectory_containers(force_kernel, kind, n, B):
    """iLQR(env, storage_bf16=True) on a shared env: the 16-per-wave kernel keeps BOTH trajectory buffers of a column as
    real bf16 arrays in the workspace (half the bytes of every pass) and widens the final nominal trajectory into the
    fp32 outputs.  Definition of the mode (TfmpcIlqrConfig::storage_bf16): values are rounded to nearest even when stored,
    arithmetic stays fp32 -- what the wave kernel emulates in fp32 containers.  Checked: outputs are bf16-representable;
    the start rollout equals the fp32 one rounded; after one iteration the two implementations of the format agree to a
    few bf16 ulps on most instances (they round different intermediate quantities: the wave kernel stores the gain k,
    this kernel one selector bit and rebuilds k from the rounded u)."""
    T = 25
    env, x0 = _env(kind, n, B, 3)
    u0 = iLQR(env).random_actions(T, B, seed=4)

    def run(kern, **kw):
        force_kernel(kern)
        out = iLQR(env, **kw).solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
        return out

    start16 = run("costate_mfma", max_iterations=1, atol=1e9, storage_bf16=True)
    start32 = run("costate_mfma", max_iterations=1, atol=1e9)
    for key in ("states", "costs"):
        bits = start16[key].contiguous().view(torch.int32)
        assert int((bits & 0xFFFF).abs().sum()) == 0, key                      # bf16-representable
        rounded = (start32[key].contiguous().view(torch.int32) + 0x7FFF + ((start32[key].contiguous().view(torch.int32) >> 16) & 1)) & ~0xFFFF
        assert torch.equal(bits, rounded), key                                 # == the fp32 rollout, rounded to nearest even
    one16 = run("costate_mfma", max_iterations=1, storage_bf16=True)
    emu = run("wave", max_iterations=1, storage_bf16=True)
    assert int((one16["actions"].contiguous().view(torch.int32) & 0xFFFF).abs().sum()) == 0
    rel = ((one16["states"] - emu["states"]).abs().flatten(1).amax(dim=1) / emu["states"].abs().flatten(1).amax(dim=1)).cpu().numpy()
    assert np.median(rel) <= 2e-2 and np.quantile(rel, 0.9) <= 0.1, (np.median(rel), np.quantile(rel, 0.9))
    full = run(None, max_iterations=6, storage_bf16=True)
    ref = run(None, max_iterations=6)
    tf_, tr = full["costs"].sum(dim=1), ref["costs"].sum(dim=1)
    assert bool(torch.isfinite(tf_).all())
    # a perturbation of the solve, not a different one (Reservoir is bang-bang: its line-search decisions flip under a
    # perturbation of any size, so individual solves move by tens of per cent at equal quality)
    assert float(((tf_ - tr).abs() / tr.abs()).median()) <= (0.05 if kind == "hvac" else 0.3)


@pytest.mark.parametrize("kind,n,T,B", [("reservoir", 32, 37, 70), ("hvac", 32, 37, 70), ("reservoir", 4, 50, 300), ("hvac", 6, 50, 150),
                                        ("reservoir", 12, 3, 40), ("hvac", 16, 9, 33), ("reservoir", 8, 100, 64)])
def test_every_group_form_returns_the_same_bits(force_kernel, kind, n, T, B):
    """Waves per sixteen-column group (TFMPC_COSTATE_WAVES = 1 | 2 | 4 | 8; the launcher picks by batch size): one step size
    per wave in a line-search pass, the stored rollout of the accepted one as one SEGMENT of the horizon per wave from the
    accepted chain's checkpoints.  Same expressions on the same inputs in every form: identical bits, traces included
    (also for horizons shorter than the number of waves and not divisible by it)."""
    env, x0 = _env(kind, n, B, n + 1)
    solver = iLQR(env, max_iterations=5)
    u0 = solver.random_actions(T, B, seed=n)
    force_kernel("costate_mfma")
    outs = {}
    for waves in ("1", "2", "4", "8"):
        with _hip.option("TFMPC_COSTATE_WAVES", waves):
            o = solver.solve_device(x0, T, u_init=u0, trace_rows=8)
            torch.cuda.synchronize()
            outs[waves] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}
    def decisions(o):        # (the J of a REJECTED last try is a partial sum up to where its pass was cut short: that depends on the form)
        tr = o["trace"].clone()
        tr[..., 7] = torch.where(tr[..., 8] > 0, tr[..., 7], torch.zeros_like(tr[..., 7]))
        rows = torch.arange(tr.shape[1], device=tr.device)[None, :, None] < o["trace_len"][:, None, None]
        return torch.where(rows, tr, torch.zeros_like(tr))
    # HVAC: the multi-wave forms keep part of the matrix operand in LDS and add the six partial products in another order
    # than the one-wave form (fp32 rounding: a decision can flip): identical among themselves, close to the one-wave form
    base = "2" if kind == "hvac" else "1"
    for waves in ("2", "4", "8"):
        for key in ("iterations", "status", "states", "actions", "costs", "trace_len"):
            assert torch.equal(outs[waves][key], outs[base][key]), (waves, key)
        assert torch.equal(decisions(outs[waves]), decisions(outs[base])), waves
    if kind == "hvac":
        c1, c2 = outs["1"]["costs"].sum(dim=1), outs["2"]["costs"].sum(dim=1)
        assert int(((c1 - c2).abs() > 1e-4 * c1.abs()).sum()) <= max(1, B // 10)


@pytest.mark.parametrize("n,T,B", [(32, 100, 64), (20, 40, 33), (17, 12, 5), (31, 25, 16),
                                   (4, 100, 48), (3, 30, 21), (8, 40, 30), (6, 25, 17), (7, 12, 9), (16, 50, 20), (12, 33, 40), (2, 9, 5)])
def test_shift_coupling_is_the_matrix_product_bit_for_bit(force_kernel, n, T, B):
    """Round 4: the `downstream` matrix of a chain of reservoirs (every config the reference holds:
    /root/reference/tests/conftest.py:70-75, tfmpc/envs/reservoir/res4.config.json:13-18) is a shift, and the two-tile kernel moves
    rows instead of multiplying (ilqr_adjoint_mfma.hip:shift_apply).  TFMPC_COSTATE_COUPLING=dense keeps the products: every output
    and the decision trace must be the same bits, also for n < 32 (the padding row below the last reservoir must stay 0), and for the
    one-tile kernels with 1, 2 or 4 instances per matrix-core column (n <= 16 / 8 / 4: the shift stays inside an instance's rows)."""
    env, x0 = _env("reservoir", n, B, n)
    solver = iLQR(env, max_iterations=8)
    u0 = solver.random_actions(T, B, seed=n)
    force_kernel("costate_mfma")
    out = {}
    for mode in (None, "dense"):
        with _hip.option("TFMPC_COSTATE_COUPLING", mode):
            o = solver.solve_device(x0, T, u_init=u0, trace_rows=24)
            torch.cuda.synchronize()
            out[mode] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}
    for key in ("iterations", "status", "states", "actions", "costs", "trace_len"):
        assert torch.equal(out[None][key], out["dense"][key]), key
    assert torch.equal(torch.nan_to_num(out[None]["trace"]), torch.nan_to_num(out["dense"]["trace"]))
    assert int(out[None]["iterations"].max()) >= 2 and bool(torch.isfinite(out[None]["costs"]).all())


@pytest.mark.parametrize("n", [32, 22])
def test_a_downstream_matrix_that_is_no_chain_keeps_the_products(force_kernel, n):
    """Two separate chains (the link 5 -> 6 removed) are no shift of the whole state: the kernel must notice and multiply -- and with
    one nonzero per row it stays bit-identical to the wave kernel, as before.  (A tree, two reservoirs feeding one, has rows of two
    terms that the matrix cores add in another order than the wave kernel: fp32 tolerance there, as for dense couplings.)"""
    T, B = 30, 20
    cfg = dict(problems.reservoir_config(n, seed=n))
    D = np.array(cfg["downstream"])
    D[5, 6] = 0.0
    cfg["downstream"] = D.tolist()
    env = Reservoir.load(cfg)
    x0 = np.random.default_rng(n).uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=5)
    u0 = solver.random_actions(T, B, seed=n)
    out = _both(force_kernel, solver, x0, T, u0)
    for key in ("iterations", "status", "states", "actions", "costs"):
        assert torch.equal(out["costate_mfma"][key], out["wave"][key]), key
    # ... and it is NOT what the unbroken chain gives (the removed link matters)
    chain = iLQR(Reservoir.load(dict(problems.reservoir_config(n, seed=n))), max_iterations=5)
    force_kernel("costate_mfma")
    assert not torch.equal(chain.solve_device(x0, T, u_init=u0)["states"], out["costate_mfma"]["states"])


@pytest.mark.parametrize("n,T,B,direction", [(32, 60, 64, 1), (32, 60, 37, -1), (20, 40, 33, 1), (17, 25, 16, -1), (31, 12, 5, 1)])
def test_compile_time_chain_instantiation_is_the_general_kernel_bit_for_bit(force_kernel, n, T, B, direction):
    """Round 5: a Reservoir env whose `downstream` is a chain says so (TfmpcEnv.coupling_shift = +1: i drains into i + 1, -1: into i - 1;
    tfmpc.envs.reservoir sets it from the config) and a one-wave two-tile launch takes the instantiation whose coupling is a row move at
    COMPILE time (ilqr_adjoint_mfma.hip: EnvM<kEnvReservoirChain>).  TFMPC_COSTATE_COUPLING=runtime keeps the general kernel (run-time shift
    test), =dense its products: all three and the wave kernel return the same bits, in both directions, also for n < 32 (padding rows)."""
    cfg = dict(problems.reservoir_config(n, seed=n))
    if direction < 0:
        cfg["downstream"] = np.array(cfg["downstream"]).T.tolist()
    env = Reservoir.load(cfg)
    assert env.coupling_shift == direction
    x0 = np.random.default_rng(n).uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=8)
    u0 = solver.random_actions(T, B, seed=n)
    force_kernel("costate_mfma")
    out = {}
    with _hip.option("TFMPC_COSTATE_WAVES", "1"):
        for mode in (None, "runtime", "dense"):
            with _hip.option("TFMPC_COSTATE_COUPLING", mode):
                o = solver.solve_device(x0, T, u_init=u0, trace_rows=24)
                torch.cuda.synchronize()
                out[mode] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}
    force_kernel("wave")
    wave = solver.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    for mode in ("runtime", "dense"):
        for key in ("iterations", "status", "states", "actions", "costs", "trace_len"):
            assert torch.equal(out[None][key], out[mode][key]), (mode, key)
        assert torch.equal(torch.nan_to_num(out[None]["trace"]), torch.nan_to_num(out[mode]["trace"]))
    for key in ("iterations", "status", "states", "actions", "costs"):
        assert torch.equal(out[None][key], wave[key]), key
    assert int(out[None]["iterations"].max()) >= 2 and bool(torch.isfinite(out[None]["costs"]).all())


def test_a_broken_chain_promise_is_refused_on_the_device(force_kernel):
    """TfmpcEnv.coupling_shift is a promise by the caller; the chain instantiation checks it against the matrix before it computes anything
    and leaves TFMPC_ST_ENV_FLAG in every instance's status.  (tfmpc.envs.reservoir derives the field from the matrix, so only a caller of
    the C ABI can get this wrong; here the field is overwritten by hand.)"""
    n, T, B = 32, 20, 24
    cfg = dict(problems.reservoir_config(n, seed=3))
    D = np.array(cfg["downstream"])
    D[5, 6] = 0.0                                                     # two chains: not a shift of the whole state
    cfg["downstream"] = D.tolist()
    env = Reservoir.load(cfg)
    assert env.coupling_shift == 0
    x0 = np.random.default_rng(1).uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)
    solver = iLQR(env, max_iterations=3)
    u0 = solver.random_actions(T, B, seed=1)
    force_kernel("costate_mfma")
    with _hip.option("TFMPC_COSTATE_WAVES", "1"):
        good = solver.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
        assert int(good["status"].abs().sum()) == 0
        env._c_env_cache = None
        env.c_env()[0].coupling_shift = 1                             # the false statement, written into the struct the C ABI receives
        liar = iLQR(env, max_iterations=3)
        bad = liar.solve_device(x0, T, u_init=u0)
        torch.cuda.synchronize()
        assert bad["status"].tolist() == [_hip.ST_ENV_FLAG] * B
        # a refused solve reads as zeros, never as uninitialised memory, and the host API raises
        assert float(bad["states"].abs().max()) == 0.0 and float(bad["costs"].abs().max()) == 0.0
        with pytest.raises(ValueError, match="coupling_shift"):
            liar.solve(x0, T, show_progress=False, u_init=u0)


def test_coupling_shift_follows_the_current_downstream_matrix():
    """`downstream` is a public attribute; the promise handed to the kernels is derived from the matrix as it is NOW."""
    env = Reservoir.load(dict(problems.reservoir_config(8, seed=3)))
    assert env.coupling_shift == 1
    D = np.array(env.downstream)
    D[2, 3] = 0.0
    env.downstream = D
    assert env.coupling_shift == 0
    env.downstream = np.eye(8, k=-1, dtype=np.float32)
    assert env.coupling_shift == -1


def test_a_diverging_instance_stays_in_its_column(force_kernel):
    """a row move equals the matrix product only for FINITE states -- the product spreads one row's Inf / NaN into every
    row of ITS column (0 x Inf), the move hands it to the neighbouring row only -- so a diverging instance may end differently in the
    chain / run-time-shift / dense forms (include/tfmpc_hip.h says so).  What must hold in every form: the instance is flagged
    (TFMPC_ST_NAN) and the fifteen instances that share its wavefront are the bits they are without it."""
    n, T, B = 32, 30, 40
    env, x0 = _env("reservoir", n, B, 5)
    x0 = x0.copy()
    bad = 21
    clean = x0.copy()
    x0[bad, 3, 0] = 3.0e38                                   # overflows in the first steps
    solver = iLQR(env, max_iterations=4)
    u0 = solver.random_actions(T, B, seed=6)
    force_kernel("costate_mfma")
    outs = {}
    with _hip.option("TFMPC_COSTATE_WAVES", "1"):
        ref = solver.solve_device(clean, T, u_init=u0)
        torch.cuda.synchronize()
        for mode in (None, "runtime", "dense"):
            with _hip.option("TFMPC_COSTATE_COUPLING", mode):
                o = solver.solve_device(x0, T, u_init=u0)
                torch.cuda.synchronize()
                outs[mode] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}
    others = [b for b in range(B) if b != bad]
    for mode, o in outs.items():
        assert int(o["status"][bad]) & _hip.ST_NAN, (mode, int(o["status"][bad]))
        for key in ("iterations", "status", "states", "actions", "costs"):
            assert torch.equal(o[key][others], ref[key][others]), (mode, key)
