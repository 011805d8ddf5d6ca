"""Multi-GPU pre-flight on ONE GPU (``-m gpu``): the path the driver's N = 2, 4, 8 runs take -- a ``torch.distributed.run``
launcher, ``dist.init_process_group("nccl")`` (RCCL), barriers around the timed steps, the pre-allocated receive buffers
and ``gather_trajectories(total=...)`` as the ONE collective of the data path -- exercised with a world
of one rank, which is all a one-GPU box can hold.  The launcher is started as a child process (never an exec of this
process, which has the GPU open)."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_through_the_launcher_and_rccl_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "4096", "--no-extra", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["status_flagged_instances"] == 0
    assert line["scaling"] == "strong" and line["config"]["global_batch"] == 4096 and line["config"]["batch_per_gpu"] == 4096
    g = line["gather"]
    assert g["backend"] == "nccl" and g["world_size"] == 1
    assert g["checked"] is True                                  # what rank 0 gathered IS its own shard, bit for bit
    assert line["gathered_states_shape"] == [4096, 51, 16, 1]
    assert g["bytes_per_rank"] == 4096 * (1267 + 1) * 4              # states + actions + costs per instance, + its status (bit-cast int32)
    assert "gather_error" not in line
