"""World-size-2 CPU (gloo) test of the multi-GPU plumbing: the batch is sharded
contiguously over ranks with no data-path collective, and the result trajectories are
gathered with ONE collective at the end.  The per-rank "solve" here is a
deterministic stand-in (no GPU in this container): the test covers sharding + gather."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tfmpc import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_solve(x0, T, m):
    """Deterministic per-instance 'trajectory' depending only on that instance's x0."""
    B, n = x0.shape
    t = torch.arange(T + 1, dtype=torch.float32).view(1, T + 1, 1)
    states = (x0.unsqueeze(1) * (1.0 + 0.1 * t)).unsqueeze(-1)
    actions = states[:, :T, :m] * -0.5
    costs = states[..., 0].pow(2).sum(-1)
    return states, actions, costs


class _CountingCollectives:
    """Counts the collectives a call issues (the single-gather contract of SURVEY.md 8e is checked, not assumed)."""

    def __init__(self):
        self.calls = []

    def __enter__(self):
        self.saved = {name: getattr(dist, name) for name in ("gather", "all_gather", "all_reduce", "broadcast", "barrier")}
        for name, fn in self.saved.items():
            setattr(dist, name, (lambda fn_, name_: lambda *a, **k: (self.calls.append(name_), fn_(*a, **k))[1])(fn, name))
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(dist, name, fn)
        return False


def _worker(rank, world, port, B, n, m, T, out_path, with_total=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x0 = torch.from_numpy(np.random.default_rng(0).normal(size=(B, n)).astype(np.float32))
        mine = parallel.shard(x0)
        lo, hi = parallel.shard_bounds(B, world, rank)
        assert mine.shape[0] == hi - lo
        if with_total:
            shards = _fake_solve(mine, T, m)
            recv = parallel.gather_buffers(*shards, total=B)
            assert (recv is not None) == (rank == 0)
            with _CountingCollectives() as counted:
                res = parallel.gather_trajectories(*shards, total=B, recv=recv)
            assert counted.calls == ["gather"], counted.calls          # ONE collective, nothing else
            parallel.check_shard_sizes(mine.shape[0], B)                # start-up agreement: passes on both ranks
            try:                                                       # ... and fails on BOTH when one rank's shard is wrong
                parallel.check_shard_sizes(mine.shape[0] + (1 if rank == 1 else 0), B)
                raise AssertionError("a wrong shard on rank 1 went unnoticed on rank %d" % rank)
            except ValueError:
                pass
            try:                                                       # a shard of the wrong size is refused BEFORE any collective
                with _CountingCollectives() as counted:
                    parallel.gather_trajectories(*_fake_solve(torch.cat([mine, x0[:1]]), T, m), total=B)
                raise AssertionError("a shard that is not the block split's was accepted")
            except ValueError:
                assert counted.calls == []
        else:
            res = parallel.gather_trajectories(*_fake_solve(mine, T, m))
        if rank == 0:
            full = _fake_solve(x0, T, m)
            ok = all(torch.equal(a, b) for a, b in zip(res, full))
            torch.save({"ok": ok, "shapes": [tuple(r.shape) for r in res]}, out_path)
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_the_batch_contiguously():
    for total in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            bounds = [parallel.shard_bounds(total, world, r) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == total
            assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in bounds]
            assert max(sizes) - min(sizes) <= 1


def test_gather_is_identity_without_process_group():
    s, a, c = _fake_solve(torch.ones(3, 4), 5, 2)
    out = parallel.gather_trajectories(s, a, c)
    assert out[0] is s and out[1] is a and out[2] is c


def test_two_rank_shard_and_gather(tmp_path):
    B, n, m, T = 37, 6, 3, 9           # odd batch: ranks get 19 and 18 instances
    out_path = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, _free_port(), B, n, m, T, out_path), nprocs=2, join=True)
    res = torch.load(out_path)
    assert res["ok"]
    assert res["shapes"] == [(B, T + 1, n, 1), (B, T, m, 1), (B, T + 1)]


def test_two_rank_gather_with_an_empty_shard(tmp_path):
    B, n, m, T = 1, 4, 2, 5            # fewer instances than ranks: rank 1 owns nothing
    out_path = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, _free_port(), B, n, m, T, out_path), nprocs=2, join=True)
    res = torch.load(out_path)
    assert res["ok"] and res["shapes"][0] == (B, T + 1, n, 1)


def test_two_rank_gather_is_one_collective_when_the_global_batch_is_known(tmp_path):
    """`total=`: the shard sizes follow from the block split, so the data path is literally ONE dist.gather -- no size
    exchange, no agreement step (counted) -- also with shards that differ by one instance and with an empty shard."""
    for B in (37, 1, 64):
        out_path = str(tmp_path / f"rank0_{B}.pt")
        mp.spawn(_worker, args=(2, _free_port(), B, 6, 3, 9, out_path, True), nprocs=2, join=True)
        res = torch.load(out_path)
        assert res["ok"] and res["shapes"][0] == (B, 10, 6, 1)


# ---- round 6: what iLQR.solve returns beside the trajectory rides in the same gather (iterations[B/G], status[B/G]) ----

def _fake_ilqr_out(x0, T, m):
    """Stand-in for iLQR.solve_device on a machine without a GPU: a deterministic 'solution' per instance, with int32 columns whose bit
    patterns are NaNs / denormals when read as fp32 (a gather that did arithmetic on the packed buffer would damage them)."""
    states, actions, costs = _fake_solve(x0, T, m)
    key = (x0[:, 0] * 1000.0).to(torch.int32)
    iterations = key.abs() % 101
    status = torch.where(key % 3 == 0, torch.full_like(key, 0x7fc00001), key % 64).to(torch.int32)      # 0x7fc00001: a NaN pattern
    return dict(states=states, actions=actions, costs=costs, iterations=iterations, status=status, batched=True, workspace=None)


def _ilqr_worker(rank, world, port, B, n, m, T, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tfmpc.envs.lqr.navigation import NavigationLQR
        from tfmpc.solvers.ilqr import iLQR
        x0 = torch.from_numpy(np.random.default_rng(0).normal(size=(B, n)).astype(np.float32))
        mine = parallel.shard(x0)
        solver = iLQR.__new__(iLQR)                         # the host class without its device (none here): solve() only orchestrates
        solver.max_iterations, solver.max_attempts = 100, 10
        solver.env = type("Env", (), {"c_env": lambda self: (type("E", (), {"coupling_shift": 0})(), [])})()
        solver.solve_device = lambda x0_, T_, u_init=None, seed=None, trace_rows=0: _fake_ilqr_out(mine, T_, m)
        with _CountingCollectives() as counted:
            traj, iterations = solver.solve(mine, T, show_progress=False, gather=True, total=B)
        assert counted.calls == ["gather"], counted.calls            # ONE collective for trajectory + iterations + status
        if rank == 0:
            full = _fake_ilqr_out(x0, T, m)
            ok = (np.array_equal(traj.states, full["states"][..., 0].numpy()) and np.array_equal(traj.actions, full["actions"][..., 0].numpy())
                  and np.array_equal(traj.costs, full["costs"].numpy()))                    # (Trajectory squeezes the column axis)
            ok = ok and iterations.dtype == np.int32 and np.array_equal(iterations, full["iterations"].numpy())
            ok = ok and solver.last_status.dtype == torch.int32 and torch.equal(solver.last_status, full["status"])
            torch.save({"ok": bool(ok), "len": len(iterations)}, out_path)
        else:
            assert traj is None and iterations is None
        # the general form (sizes not known up front) carries the columns too
        out = _fake_ilqr_out(mine, T, m)
        res = parallel.gather_results(out["states"], out["actions"], out["costs"], iterations=out["iterations"], status=out["status"])
        if rank == 0:
            assert torch.equal(res[3], _fake_ilqr_out(x0, T, m)["iterations"]) and torch.equal(res[4], _fake_ilqr_out(x0, T, m)["status"])
        # receive buffers sized without the int columns are refused before anything is sent (rank 0 only holds buffers)
        if rank == 0:
            short = parallel.gather_buffers(out["states"], out["actions"], out["costs"], total=B)
            try:
                parallel.gather_results(out["states"], out["actions"], out["costs"], iterations=out["iterations"], status=out["status"],
                                        total=B, recv=short)
                raise AssertionError("receive buffers without the int columns were accepted")
            except ValueError:
                pass
    finally:
        dist.destroy_process_group()


def test_two_rank_ilqr_solve_gathers_trajectory_iterations_and_status_in_one_collective(tmp_path):
    for B in (37, 1):
        out_path = str(tmp_path / f"ilqr_{B}.pt")
        mp.spawn(_ilqr_worker, args=(2, _free_port(), B, 6, 3, 9, out_path), nprocs=2, join=True)
        res = torch.load(out_path)
        assert res["ok"] and res["len"] == B


def test_gather_results_refuses_columns_that_are_not_int32_per_instance():
    s, a, c = _fake_solve(torch.ones(3, 4), 5, 2)
    import pytest
    with pytest.raises(ValueError):
        parallel.gather_results(s, a, c, iterations=torch.zeros(3, dtype=torch.int64))
    with pytest.raises(ValueError):
        parallel.gather_results(s, a, c, status=torch.zeros(4, dtype=torch.int32))
    out = parallel.gather_results(s, a, c, iterations=torch.arange(3, dtype=torch.int32))      # no process group: identity
    assert out[0] is s and out[3].tolist() == [0, 1, 2] and out[4] is None


def _build_worker(idx, source, out_dir):
    os.environ["TFMPC_USERENV_CACHE"] = out_dir
    from tfmpc.envs import deviceenv
    deviceenv._CACHE = os.path.join(out_dir, "primary")            # a fresh cache: every process finds nothing and compiles
    path = deviceenv.build(source, 2, 2, 8)
    with open(os.path.join(out_dir, f"path{idx}.txt"), "w") as fh:
        fh.write(path)


def test_ranks_compiling_the_same_device_env_at_once_end_with_one_valid_library(tmp_path):
    """N ranks of a job reach their first DeviceEnv solve together: each compiles into its own temporary and os.replace()s it onto the
    same name -- every rank must come back with the same path and a library that loads and exports the twins."""
    import ctypes
    import sys
    from tfmpc.envs import deviceenv
    if deviceenv.hipcc_path() is None:
        import pytest
        pytest.skip("no hipcc in this environment")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import deviceenv_sources as sources
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_build_worker, args=(i, sources.NAVIGATION, str(tmp_path))) for i in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    paths = {open(os.path.join(str(tmp_path), f"path{i}.txt")).read() for i in range(4)}
    assert len(paths) == 1
    path = paths.pop()
    lib = ctypes.CDLL(path)
    for twin in deviceenv._TWINS.values():
        assert hasattr(lib, twin), twin
    leftovers = [f for f in os.listdir(os.path.dirname(path)) if f.endswith(".tmp") or (f.startswith("env.") and f != "env.hip")]
    assert leftovers == [], leftovers
