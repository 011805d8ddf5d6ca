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
