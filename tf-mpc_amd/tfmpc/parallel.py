"""Multi-GPU use of the path: problem instances are independent, so the batch is
sharded over one process per GPU with NO collective during the solve; the only
exchange is ONE gather of the result trajectories at the end (RCCL over xGMI when
the tensors are on GPUs, gloo on CPU in the tests).  SURVEY.md §8(e)."""

import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous block split of ``total`` instances over ``world_size`` ranks."""
    base, rem = divmod(int(total), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard(tensor, world_size=None, rank=None):
    """This rank's contiguous block of a batch-major tensor."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(tensor.shape[0], world_size, rank)
    return tensor[lo:hi]


def check_shard_sizes(b, total, group=None):
    """Start-up agreement (NOT part of the data path): every rank checks that the ``b`` instances it holds are what the
    block split of ``total`` gives it, and the ranks all_reduce a 4-byte flag, so a wrong shard raises ``ValueError`` on
    EVERY rank.  ``gather_trajectories(total=...)`` itself stays one collective: there a wrong shard raises on its own rank
    only, before anything is sent, and the other ranks would wait inside the gather -- call this once when the job starts."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(total, world, rank)
    ok = torch.tensor([1 if int(b) == hi - lo else 0], dtype=torch.int32,
                      device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        raise ValueError(f"check_shard_sizes: at least one rank holds a shard that is not the block split of {total} "
                         f"(this rank: {int(b)} instances, the split gives it {hi - lo})")


def _row_floats(states, actions, costs, ints):
    return states.shape[1:].numel() + actions.shape[1:].numel() + costs.shape[1:].numel() + len(ints)


def gather_buffers(states, actions, costs, total, dst=0, group=None, iterations=None, status=None):
    """The receive buffers of ``gather_results(..., total=total)`` on rank ``dst`` (``None`` elsewhere): allocate
    them when the job starts, where running out of memory is an ordinary start-up error, instead of in front of the
    collective.  Pass ``iterations`` / ``status`` iff the gather will carry them (one more float per row each)."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    world = dist.get_world_size(group)
    if dist.get_rank() != dst:
        return None
    per = _row_floats(states, actions, costs, [t for t in (iterations, status) if t is not None])
    bmax = max(hi - lo for lo, hi in (shard_bounds(total, world, r) for r in range(world)))
    return [torch.empty((bmax, per), device=states.device, dtype=states.dtype) for _ in range(world)]


def gather_results(states, actions, costs, iterations=None, status=None, dst=0, group=None, total=None, recv=None):
    """Gather per-rank result shards ``states[b,T+1,n,1]``, ``actions[b,T,m,1]``, ``costs[b,T+1,...]`` and -- what
    ``iLQR.solve`` returns beside the trajectory (ilqr.py:281-283) -- ``iterations[b]``, ``status[b]`` (int32) on GLOBAL
    rank ``dst`` as ONE collective over a single packed fp32 buffer: the int32 columns ride in it BIT-CAST (a gather moves
    bytes, no arithmetic touches them).  Returns ``(states, actions, costs, iterations, status)`` on ``dst`` (the last two
    ``None`` when not given) and ``None`` elsewhere.  SURVEY.md 8(e)'s single RCCL gather.

    ``total`` = the global number of instances of a block-sharded batch (``shard`` / ``shard_bounds``): every rank
    then knows every shard's size, and the ONLY communication is one ``dist.gather`` of the packed rows (padded to
    the largest shard, which is at most one row more than the smallest).
    ``recv`` = buffers from ``gather_buffers`` (else they are allocated here; an allocation failure then raises on
    that rank only).  A shard whose size is not the block split's raises ``ValueError`` on ITS rank before anything is
    sent -- the other ranks would then wait in the gather: ``check_shard_sizes`` at start-up makes that error collective.

    Without ``total`` the shards may be of any sizes: an 8-byte ``all_gather`` of the sizes and a 4-byte
    ``all_reduce`` by which the ranks agree that every buffer could be allocated precede the gather, so that an
    out-of-memory raises ``RuntimeError`` on EVERY rank instead of leaving the others inside the collective."""
    ints = [t for t in (iterations, status) if t is not None]
    for t in ints:
        if t.dtype != torch.int32 or t.shape != (states.shape[0],):
            raise ValueError("iterations / status: int32 [b], one entry per instance of this rank's shard")
    if states.dtype != torch.float32 and ints:
        raise ValueError("the int32 columns are bit-cast into an fp32 buffer: trajectories must be float32")
    if not (dist.is_available() and dist.is_initialized()):
        return states, actions, costs, iterations, status
    world = dist.get_world_size(group)
    is_dst = dist.get_rank() == dst                          # `dst` is a global rank, as dist.gather takes it
    b = states.shape[0]
    ns, na, nc = states.shape[1:].numel(), actions.shape[1:].numel(), costs.shape[1:].numel()   # valid for an empty shard too
    per = ns + na + nc + len(ints)

    def pack(bmax):
        packed = torch.zeros((bmax, per), device=states.device, dtype=states.dtype)
        cols = [states.reshape(b, ns), actions.reshape(b, na), costs.reshape(b, nc)]
        cols += [t.to(states.device).contiguous().view(torch.float32).reshape(b, 1) for t in ints]      # same bits, fp32 label
        packed[:b] = torch.cat(cols, dim=1)
        return packed

    def unpack(recv_, all_b):
        full = torch.cat([r[:nb] for r, nb in zip(recv_, all_b)], dim=0)
        B = full.shape[0]
        out = [full[:, :ns].reshape(B, *states.shape[1:]), full[:, ns:ns + na].reshape(B, *actions.shape[1:]),
               full[:, ns + na:ns + na + nc].reshape(B, *costs.shape[1:])]
        col = ns + na + nc
        for t in (iterations, status):
            if t is None:
                out.append(None)
            else:
                out.append(full[:, col].contiguous().view(torch.int32))
                col += 1
        return tuple(out)

    if total is not None:
        all_b = [hi - lo for lo, hi in (shard_bounds(total, world, r) for r in range(world))]
        mine = all_b[dist.get_rank(group)]
        if b != mine:                                        # a caller's error, raised before anything is sent
            raise ValueError(f"gather_results: this rank holds {b} instances, the block split of {total} gives it {mine}")
        packed = pack(max(all_b))
        if is_dst and recv is None:
            recv = [torch.empty_like(packed) for _ in range(world)]
        if is_dst and tuple(recv[0].shape) != tuple(packed.shape):
            raise ValueError(f"gather_results: receive buffers are {tuple(recv[0].shape)}, the packed rows {tuple(packed.shape)} "
                             "(gather_buffers must be told about iterations / status)")
        dist.gather(packed, recv if is_dst else None, dst=dst, group=group)          # the one collective
        return unpack(recv, all_b) if is_dst else None

    sizes = torch.tensor([b], device=states.device, dtype=torch.int64)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)          # 8 bytes per rank; sizes only
    all_b = [int(s.item()) for s in all_sizes]
    packed, error = None, None
    try:
        packed = pack(max(all_b))
        recv = [torch.empty_like(packed) for _ in range(world)] if is_dst else None
    except RuntimeError as exc:                              # e.g. out of memory for the receive buffers
        error = exc
    ok = torch.tensor([0 if error is not None else 1], device=states.device, dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        raise RuntimeError(f"gather_results: buffer allocation failed on at least one rank ({error!r} here)")
    dist.gather(packed, recv, dst=dst, group=group)        # the one data-path collective
    return unpack(recv, all_b) if is_dst else None


def gather_trajectories(states, actions, costs, dst=0, group=None, total=None, recv=None):
    """``gather_results`` for the trajectory alone (what ``LQR.solve`` returns, lqr.py:163-166): ``(states, actions, costs)`` on
    ``dst``, ``None`` elsewhere, the inputs themselves without a process group."""
    res = gather_results(states, actions, costs, dst=dst, group=group, total=total, recv=recv)
    return None if res is None else res[:3]


def gather_bytes_per_rank(states, actions, costs, iterations=None, status=None):
    """Payload one rank contributes to the gather (the packed fp32 rows)."""
    extra = sum(int(t.numel()) for t in (iterations, status) if t is not None)
    return (int(states.numel() + actions.numel() + costs.numel()) + extra) * states.element_size()
