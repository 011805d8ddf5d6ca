"""bench.py's multi-rank control flow (rank env vars, barriers, MAX over ranks, the single
final gather, ONE JSON line on rank 0) exercised on CPU over gloo with --dry-run-cpu, launched
exactly as the driver launches the real run (torch.distributed.run, 127.0.0.1)."""

import json
import os
import socket
import subprocess
import sys

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout           # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_single_process_dry_run():
    line = _run([sys.executable, "bench.py", "--dry-run-cpu", "--steps", "3", "--warmup", "1"])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["valid"] is False


def test_two_rank_dry_run_aggregates_over_ranks():
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                 "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                 "bench.py", "--gpus", "2", "--dry-run-cpu", "--steps", "4", "--warmup", "1"])
    assert line["n_gpus"] == 2 and line["total_instances"] == 64 + 65          # whole-job aggregate
    assert line["gathered_states_shape"] == [129, 6, 4, 1] and line["gathered_rank_of_last_instance"] == 1.0
    assert line["ms_per_step"] >= 3.5                                          # the slower rank (4 ms sleeps) sets the time
