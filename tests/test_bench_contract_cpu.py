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


def test_strong_and_weak_scaling_modes_shard_differently():
    """--scaling strong (the default: BASELINE configs[2], a fixed global batch block-sharded over the ranks) against
    --scaling weak (a fixed batch per rank), three ranks: the dry run's global batch is 129 = 43 + 43 + 43 (strong) or
    64 x 3 + 1 = 193 = 65 + 64 + 64 (weak); the line says which mode it ran."""
    def run(mode):
        return _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                     "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                     "bench.py", "--gpus", "3", "--dry-run-cpu", "--steps", "2", "--warmup", "1"] + (["--scaling", mode] if mode else []))
    strong, weak = run(None), run("weak")
    assert strong["scaling"] == "strong" and strong["total_instances"] == 129 and strong["gathered_states_shape"][0] == 129
    assert weak["scaling"] == "weak" and weak["total_instances"] == 193 and weak["gathered_states_shape"][0] == 193


def test_gpus_flag_alone_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher (the form the driver uses for N = 1) must not silently run one
    rank: bench.py starts the two ranks as a torch.distributed.run child process and rank 0 prints the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run-cpu", "--steps", "2", "--warmup", "1"],
                         cwd=ROOT, env=dict(env, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["total_instances"] == 64 + 65


def test_gpus_flag_must_match_the_launcher():
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--dry-run-cpu"], cwd=ROOT,
                         env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_secondary_numbers_come_from_a_child_that_cannot_cost_the_headline_line():
    """bench.py measures `extra` in a child process started BEFORE the parent touches the GPU and released over stdin
    after the headline measurement; a child that dies, prints junk or hangs becomes an `error` entry."""
    sys.path.insert(0, ROOT)
    import bench

    def child(code):
        return subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True)

    good = bench.collect_extras(child("import sys, json; assert sys.stdin.readline() == 'go\\n'; print('noise'); print(json.dumps({'a': 1}))"))
    assert good == {"a": 1}
    died = bench.collect_extras(child("import sys; sys.stdin.readline(); sys.stderr.write('boom'); sys.exit(3)"))
    assert "exited with 3" in died["error"] and "boom" in died["stderr_tail"]
    junk = bench.collect_extras(child("import sys; sys.stdin.readline(); print('not json')"))
    assert "error" in junk
    hung = bench.collect_extras(child("import sys, time; sys.stdin.readline(); time.sleep(60)"), timeout=2)
    assert "did not finish" in hung["error"]
    # the child side leaves quietly when the parent goes away before releasing it
    out = subprocess.run([sys.executable, "bench.py", "--extras-only"], cwd=ROOT, stdin=subprocess.DEVNULL,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 1 and out.stdout.strip() == ""


def test_extra_summary_is_compact_and_survives_missing_entries():
    """`extra_summary` repeats the secondary numbers as the LAST key of the line (a log that keeps only the tail of the line still shows
    them); it must never raise, whatever `extra` holds (an `error` entry, a missing workload)."""
    sys.path.insert(0, ROOT)
    import bench
    empty = bench.summarise_extras({"error": "the extras process exited with 3"})
    assert empty["ilqr_api_warm"] == [None, None, None] and empty["cfg5_hvac"] == [None, None]        # (ms, frac, executed-flop frac: round 6)
    full = bench.summarise_extras({
        "ilqr_api": {"ms_per_batch": 5.4588, "roofline": {"frac": 0.35744, "frac_executed": 0.2011}, "cold_start": {"ms_per_batch": 5.57, "roofline": {"frac": 0.357}},
                     "control_limited": {"ms_per_batch": 512.6, "stable_open_loop_variant": {"ms_per_batch": 88.86}}},
        "other_configs": {"cfg5_hvac_ilqr_n32": {"ms_per_batch": 13.147, "roofline": {"frac": 0.1938}, "algorithmic_flop_rate": {"frac": 0.4294}},
                          "cfg4_navigation_ilqr": {"ms_per_batch": 9.19, "one_launch_of_8x16384_instances": {"iterations_per_s": 56.14e6}}},
        "bf16_storage_sweep": {"error": "x"}})
    assert full["ilqr_api_warm"] == [5.46, 0.357, 0.201] and full["cfg5_hvac"] == [13.1, 0.194, 0.429] and full["cfg4_one_launch_8x16384_Mit_s"] == 56.1
    assert full["bf16_sweep"] is None and len(json.dumps(full)) < 1200
