"""CPU tests of the host side of the LQR boundary: the C-ABI library loads and
exports every symbol ``include/*.h`` declares, argument handling, containers.
No compute call is made (there is no GPU here and no CPU fallback)."""

import ctypes
import glob
import io
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from tfmpc import _hip
from tfmpc.envs import make_lqr, make_lqr_linear_navigation
from tfmpc.solvers.lqr import LQR
from tfmpc.utils.trajectory import Trajectory


def _declared_symbols():
    names = []
    for header in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
        names += re.findall(r"\b(tfmpc_\w+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_hip.lib_path())
    declared = _declared_symbols()
    assert len(declared) >= 6
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    for name in _hip._SIGNATURES:
        assert name in declared, f"{name} bound in _hip.py but not declared in include/"


def test_kernel_choice_and_workspace_queries():
    lib = _hip.load()
    assert lib.tfmpc_version() >= 100
    assert lib.tfmpc_lqr_kernel_name(3, 2, 10).startswith(b"lane")      # tiny: lane-per-instance (big batches)
    assert lib.tfmpc_lqr_kernel_name(16, 8, 50) == b"mfma_16x8"        # BASELINE headline shape
    assert lib.tfmpc_lqr_kernel_name(12, 6, 50).startswith(b"mfma_16x8")   # zero-padded into the same tiles
    assert lib.tfmpc_lqr_kernel_name(32, 32, 10) == b"block_mfma_f32"     # large: a workgroup per instance
    assert lib.tfmpc_lqr_kernel_name(32, 16, 10) == b"mfma_32x16"          # configs[4]'s literal dims: 2 x 2 tiles of bf16x3
    assert lib.tfmpc_lqr_kernel_name(17, 2, 10).startswith(b"mfma_32x16")  # ... zero-padded
    assert lib.tfmpc_lqr_kernel_name(5, 19, 10).endswith(b"generic_wave")  # in between: by batch size
    assert lib.tfmpc_lqr_kernel_name(200, 200, 10) == b"unsupported"
    assert lib.tfmpc_lqr_workspace_bytes(4, 16, 8, 50) == 4 * 50 * 8 * 17 * 4


def test_make_lqr_shapes_and_spd():
    """reference tests/test_lqr.py:18-42"""
    np.random.seed(3)
    n, m = np.random.randint(2, 10), np.random.randint(2, 10)
    lqr = make_lqr(n, m)
    d = n + m
    assert lqr.state_size == n and lqr.action_size == m and lqr.n_dim == d
    assert lqr.F.shape == (n, d) and lqr.f.shape == (n, 1) and lqr.C.shape == (d, d) and lqr.c.shape == (d, 1)
    assert lqr.F.dtype == torch.float32
    C = lqr.C.cpu().numpy()
    assert np.allclose(C.T, C, atol=1e-2)
    np.linalg.cholesky(C)


def test_batched_and_shared_operands():
    lqr = make_lqr_linear_navigation(np.zeros((7, 2, 1)), 5.0)
    assert lqr.batch_size == 7
    assert [s for _, s in lqr._operands()] == [0, 0, 0, 4]
    with pytest.raises(ValueError):
        LQR(np.zeros((2, 3, 5)), np.zeros((4, 3, 1)), np.zeros((5, 5)), np.zeros((5, 1)))
    with pytest.raises(ValueError):
        LQR(np.zeros((3, 3)), np.zeros((3, 1)), np.zeros((3, 3)), np.zeros((3, 1)))


def test_dump_and_load_roundtrip():
    """reference tests/test_lqr.py:97-107"""
    np.random.seed(0)
    lqr = make_lqr(4, 3)
    buf = io.StringIO()
    lqr.dump(buf)
    buf.seek(0)
    lqr2 = LQR.load(buf)
    for name in ("F", "f", "C", "c"):
        assert torch.equal(getattr(lqr, name), getattr(lqr2, name))


def test_single_step_model_matches_oracle():
    from oracle import lqr_ref
    np.random.seed(1)
    lqr = make_lqr(3, 2, device="cpu")
    F, f, C, c = (t.numpy().astype(np.float64) for t in (lqr.F, lqr.f, lqr.C, lqr.c))
    x, u = np.random.normal(size=(3, 1)), np.random.normal(size=(2, 1))
    assert np.allclose(lqr.transition(x, u).numpy(), lqr_ref.transition(F, f, x, u), atol=1e-5)
    assert np.allclose(lqr.cost(x, u).numpy(), lqr_ref.cost(C, c, x, u), atol=1e-4)
    assert np.allclose(lqr.final_cost(x).numpy(), lqr_ref.final_cost(C, c, x), atol=1e-4)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_solver_fails_loudly_without_gpu():
    np.random.seed(0)
    lqr = make_lqr(3, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lqr.solve(np.zeros((3, 1)), 5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lqr.backward(5)


def test_trajectory_aggregates():
    """reference tests/test_trajectory.py:20-54"""
    rng = np.random.default_rng(0)
    T, n, m = 10, 3, 2
    states, actions = rng.normal(size=(T + 1, n, 1)), rng.normal(size=(T, m, 1))
    costs = rng.normal(size=(T + 1, 1, 1))
    traj = Trajectory(torch.as_tensor(states), actions, costs)
    assert len(traj) == T
    assert traj.states.shape == (T + 1, n) and traj.actions.shape == (T, m) and traj.costs.shape == (T + 1,)
    assert np.allclose(traj.initial_state, states[0, :, 0]) and np.allclose(traj.final_state, states[-1, :, 0])
    c = costs.reshape(-1)
    assert np.isclose(traj.total_cost, c.sum())
    assert np.allclose(traj.cumulative_cost, np.cumsum(c))
    assert np.allclose(traj.cost_to_go, np.cumsum(c[::-1])[::-1])
    s, a, cc = traj[3]
    assert np.allclose(s, states[4, :, 0]) and np.allclose(a, actions[3, :, 0]) and np.isclose(cc, c[3])
    assert "Steps" in str(traj) and "total=" in repr(traj)
    assert len(list(traj)) == T
    batch = Trajectory(np.stack([states] * 4), np.stack([actions] * 4), np.stack([costs] * 4))
    assert batch.batched and batch.total_cost.shape == (4,) and np.allclose(batch.instance(2).states, traj.states)


def test_trajectory_save_csv(tmp_path):
    rng = np.random.default_rng(0)
    traj = Trajectory(rng.normal(size=(5, 2, 1)), rng.normal(size=(4, 2, 1)), rng.normal(size=(5,)))
    path = tmp_path / "out" / "data.csv"
    traj.save(str(path))
    import pandas as pd
    df = pd.read_csv(path)
    assert list(df.columns) == ["Timestep", "x[1]", "x[2]", "u[1]", "u[2]", "costs"]
    assert len(df) == 4 and np.allclose(df["x[1]"], traj.states[1:, 0]) and np.allclose(df["costs"], traj.costs[:-1])


@pytest.mark.parametrize("name,x0", [("readme_lqr_table", [-1.0, 0.5, 3.6]), ("readme_navlin_table", [0.0, 0.0])])
def test_trajectory_str_and_repr_reproduce_the_readme_tables(name, x0):
    """The reference's README shows the exact text of ``repr(trajectory)`` and ``print(trajectory)`` for its two CLI
    examples (``/root/reference/README.md:35-51`` lqr, ``:75-91`` navlin; kept as output fixtures under
    tests/golden/).  The numbers in them are not reproducible (unseeded problem; zero-terminal older version,
    SURVEY.md F3) but the FORMAT is the interface (``trajectory.py:43-69``): a Trajectory built from the table's
    own numbers must print the table back byte for byte."""
    text = open(os.path.join(GOLDEN, name + ".txt")).read()
    header, _, table = text.partition("\n\n")
    rows = [line for line in table.splitlines()[2:] if line.strip()]
    nums = lambda cell: [float(v) for v in cell.strip(" []").split(",")]
    states = np.array([x0] + [nums(r.split("|")[1]) for r in rows], dtype=np.float32)
    actions = np.array([nums(r.split("|")[2]) for r in rows], dtype=np.float32)
    # the README examples predate the final cost (their total is the sum of the T stage costs): it is 0 here
    costs = np.array([float(r.split("|")[3]) for r in rows] + [0.0], dtype=np.float32)
    traj = Trajectory(states[..., None], actions[..., None], costs)
    assert str(traj) == table
    # repr: numpy prints the fp32 init / final vectors; the README's final state carries digits the table rounds
    # away, so compare structure and the 4-decimal total
    m = re.fullmatch(r"Trajectory\(init=\[(.*)\], final=\[(.*)\], total=(-?\d+\.\d{4})\)", header.strip())
    r = re.fullmatch(r"Trajectory\(init=\[(.*)\], final=\[(.*)\], total=(-?\d+\.\d{4})\)", repr(traj))
    assert m and r
    assert r.group(1) == m.group(1)                                        # "-1.   0.5  3.6" / "0. 0."
    assert np.allclose(np.array(r.group(2).split(), float), np.array(m.group(2).split(), float), atol=1e-4)
    assert abs(float(r.group(3)) - float(m.group(3))) <= 2e-4


def test_trace_log_lines_follow_the_reference_log_messages():
    """`trace_log_lines`: the per-pass lines of the reference's trace.log (ilqr.py:229-257, 301-303, 332-333) from decision records."""
    from tfmpc.solvers.ilqr import trace_log_lines
    records = [dict(iteration=0, mu=0.0, delta=0.0, J_hat=12.5, g_norm=0.4, alpha_index=2, alpha=0.25, J=11.0, accepted=True, residual=1.5),
               dict(iteration=1, mu=0.0, delta=0.0, J_hat=11.0, g_norm=1e-5, alpha_index=None, alpha=None, J=None, accepted=None,
                    residual=None)]
    lines = trace_log_lines(records)
    assert lines[0] == "[SOLVE] >>>>>>> Iteration = 0 <<<<<<<" and lines[4] == "[SOLVE] >>>>>>> Iteration = 1 <<<<<<<"
    assert lines[3].startswith("[FORWARD] num_iter = 3, alpha = 0.25, J = 11.0000") and lines[3].endswith("accept = True")
    assert lines[-1] == "[SOLVE] CONVERGED: g_norm < atol" and len(lines) == 8
