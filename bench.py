#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched LQR/iLQR hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
random LQR, state_dim=16, action_dim=8, horizon=50, fp32, batch=65 536 independent
instances.  `--scaling strong` (the default; BASELINE configs[2] literally: "batch=65 536,
1->8 GPUs sharded"): the GLOBAL batch is 65 536, block-sharded over the N ranks
(tfmpc.parallel.shard_bounds), so at N = 8 a GPU solves 8 192 instances per step.
`--scaling weak`: 65 536 instances PER GPU.  At N = 1 the two are the same run.  One
"step" = one pass of the hot path over the batch = one kernel launch per rank (Riccati
backward sweep + closed-loop rollout of every instance of its shard); for an LQR problem
one solve is one iLQR iteration.  Inputs are resident in HBM before the
timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher (WORLD_SIZE unset) starts the N ranks itself: a
`torch.distributed.run` CHILD process, created before this process touches the GPU.
The headline batch is drawn by tests/problems.py:make_lqr_batch_spd -- the reference's
make_lqr (tfmpc/envs/__init__.py:9-18) vectorised: F, f, c ~ N(0,1), C by make_spd_matrix's
formula (eigenvalues ~1e-3 .. n+m).

Prints ONE JSON line on rank 0 (see the task contract) with `roofline` for the
dominant kernel and `cpu_baseline` (the oracle's C port timed on this host).
"""

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

N_STATE, N_ACTION, HORIZON, BATCH = 16, 8, 50, 65536
CPU_BASELINES = True             # (--no-cpu-baseline clears it: profiling runs)
PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md:41-42 (vector == f32-MFMA dense peak)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md:36 (spec)


def lqr_flops_per_solve(n, m, T):
    """ALGORITHMIC flops of one LQR solve (multiply-add = 2 flop, no
    symmetry savings): backward 50.6 kflop/step + forward 2.2 kflop/step at n=16, m=8."""
    d = n + m
    backward = (2 * d * n * n + 2 * d * d * n + 4 * d * n) + (2.0 / 3.0 * m ** 3 + 2 * m * m * (n + 1)) \
        + (6 * n * n * m + 2 * n * m * m + 6 * n * m) + (2 * m * m + 2 * m + 2 * n * n + 2 * n)
    forward = 2 * m * n + 2 * n * d + 2 * d * d + 2 * d
    return T * (backward + forward) + 2 * n * n + 2 * n


def lqr_bytes_per_solve(n, m, T):
    """ALGORITHMIC (compulsory) HBM bytes of one solve: read F, f, C, c, x0; write
    states, actions, costs (9 132 B at n=16, m=8, T=50)."""
    d = n + m
    return 4 * (n * d + n + d * d + d + n) + 4 * ((T + 1) * n + T * m + (T + 1))


# which kernel source a profiled kernel lives in (its PMC summary is only quoted while that file is unchanged)
KERNEL_SOURCES = {"mfma": ["lqr_mfma16x8.hip", "wave_ldlt8.h", "mfma_bf16x3.h", "wave_ops.h"], "ilqr_group_solve": ["ilqr_lane.hip", "ilqr_lane_kernels.h", "envs.h"],
                  "ilqr_adjoint_mfma": ["ilqr_adjoint_mfma.hip", "trig.h"], "ilqr_lq_mfma_kernel": ["ilqr_lq_mfma.hip", "wave_ldlt8.h", "wave_ops.h"],
                  "ilqr_lq_box_mfma": ["ilqr_lq_box_mfma.hip", "wave_ldlt8.h"], "ilqr_lq_mfma32": ["ilqr_lq_mfma32.hip", "wave_ldlt.h"],
                  "lqr_mfma32x16": ["lqr_mfma32x16.hip", "wave_ldlt.h"]}


def _summary_is_current(summary, key):
    """A committed PMC summary is quoted only while the sources of its kernel hash to what it recorded
    (tools/source_stamp.py): traffic from a profile of an older kernel would be a number about other code."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import source_stamp
    return source_stamp.matches(summary.get("csrc_sha16"), KERNEL_SOURCES.get(key, []))


def measured_traffic(kernel, batch):
    """(HBM bytes per launch of the dominant kernel, where the number comes from) -- from the committed PMC passes
    (profiles/*_pmc_summary.json, collected with tools/pmc_passes.sh on this same command: separate --pmc passes,
    FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes).  bench.py cannot run rocprofv3 on itself, so the
    number is read back, scaled per instance -- and only from a summary stamped with the CURRENT kernel sources."""
    import glob
    best, best_path = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if kernel.split("_")[0] in d.get("kernel", "") and "hbm_bytes_per_launch" in d:
            best, best_path = d, path
    if best is None:
        return None, "no PMC summary under profiles/"
    if not _summary_is_current(best, "mfma"):
        return None, f"{os.path.basename(best_path)} was taken on other kernel sources (csrc_sha16 differs): not quoted"
    per_instance = best["hbm_bytes_per_launch"]["total_corrected"] / float(best.get("batch_per_launch", BATCH))
    return per_instance * batch, f"profiles/{os.path.basename(best_path)} (kernel sources unchanged since)"


def _pmc_files(file_tag=None):
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True)
    return [p for p in paths if file_tag is None or file_tag in os.path.basename(p)]


def profile_duration(kernel_substring, file_tag=None):
    """The duration of the kernel's launches in the newest committed profile taken on its current sources (tools/pmc_workload.sh: the same
    workload as the bench line, >= 40 untimed + 10 timed launches; average of the timed dispatches) -- or None."""
    for path in _pmc_files(file_tag):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        for name, k in d.get("kernel_stats", {}).items():
            if kernel_substring in name and "timed_avg_ms" in k:
                key = max((q for q in KERNEL_SOURCES if q in kernel_substring or kernel_substring in q), key=len, default=None)
                if not _summary_is_current(d, key):
                    return None
                tool = next((w["ms_per_launch"] for w in d.get("workloads", [])), None) if len(d.get("workloads", [])) == 1 else None
                return {"kernel_timed_avg_ms": k["timed_avg_ms"], "span_ms_per_launch": k.get("timed_span_ms_per_launch"), "calls": k["calls"],
                        "tool_ms_per_launch": tool, "source": "profiles/" + os.path.basename(path)}
    return None


def attach_profile(line, kernel_substring, file_tag=None):
    """Side by side: this line's ms per launch and the committed rocprofv3 duration of the same workload's kernel.  A profile whose duration
    differs from the line by more than 5 % is NOT quoted: `traffic` and `executed` of the line are withdrawn."""
    prof = profile_duration(kernel_substring, file_tag)
    if prof is None or not isinstance(line, dict) or "ms_per_batch" not in line:
        return line
    ref = prof["span_ms_per_launch"] or prof["kernel_timed_avg_ms"]
    prof["line_over_profile"] = line["ms_per_batch"] / ref
    prof["reproduces_within_5_percent"] = bool(abs(prof["line_over_profile"] - 1.0) <= 0.05)
    line["profile"] = prof
    if not prof["reproduces_within_5_percent"]:
        if isinstance(line.get("roofline"), dict):
            line["roofline"]["traffic"] = None
            line["roofline"]["traffic_withdrawn"] = "the committed profile's duration differs from this line by more than 5 %"
        for key in ("executed",):
            if key in line:
                line[key] = None
    return line


def pmc_traffic(kernel_substring, per_launch_units, units, file_tag=None):
    """HBM bytes of one launch of a secondary kernel, from the newest committed PMC summary of that kernel
    (profiles/r*_pmc.json, written by tools/pmc_kernel.sh: separate --pmc passes, FETCH_SIZE x2 + WRITE_SIZE as
    MI355X_MICROARCH.md prescribes), scaled to `units` work units (the summary was taken at `per_launch_units`).
    None when the newest summary was taken on other sources of that kernel (tools/source_stamp.py)."""
    for path in _pmc_files(file_tag):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        for name, c in d.get("counters_per_launch", {}).items():
            if kernel_substring in name and "hbm_bytes_per_launch" in c:
                if per_launch_units is None:              # (tools/pmc_workload.sh summaries say what one profiled launch processed)
                    per_launch_units = next((w.get("units_per_launch") for w in d.get("workloads", []) if w.get("units_per_launch")), None)
                    if per_launch_units is None:
                        return None
                # the LONGEST key that matches ("mfma" alone is the headline kernel and is part of every other name)
                key = max((k for k in KERNEL_SOURCES if k in kernel_substring or kernel_substring in k), key=len, default=None)
                if not _summary_is_current(d, key):
                    return None
                return c["hbm_bytes_per_launch"]["total_corrected"] * units / float(per_launch_units)
    return None


def pmc_executed(kernel_substring, file_tag=None):
    """What the kernel EXECUTED per launch, from the newest committed PMC summary taken on its current sources (else None): share of the
    SIMD cycles the vector unit / the matrix unit was busy (SQ_ACTIVE_INST_VALU x 4, SQ_VALU_MFMA_BUSY_CYCLES over 1 024 SIMDs x
    GRBM_GUI_ACTIVE / 8) and a wave's share of its residence spent in s_waitcnt -- the issue-side yardstick of the kernels whose
    algorithmic-byte or dense-flop fractions say nothing about headroom (latency- / issue-bound formulations)."""
    for path in _pmc_files(file_tag):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        for name, c in d.get("counters_per_launch", {}).items():
            if kernel_substring in name and "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
                key = max((k for k in KERNEL_SOURCES if k in kernel_substring or kernel_substring in k), key=len, default=None)
                if not _summary_is_current(d, key):
                    return None
                simd_cycles = 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0
                return {"valu_busy_frac": 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles,
                        "mfma_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles,
                        "wave_time_in_waitcnt_frac": (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
                        "valu_instructions_per_launch": c.get("SQ_INSTS_VALU"), "source": "profiles/" + os.path.basename(path)}
    return None


def ilqr_cpu_baseline(kind, cfgs, x0, u0, T, max_iterations, instances, what):
    """`cpu_baseline` of a secondary iLQR line: the fp32 numpy restatement of the reference (oracle/ilqr_ref.py: ilqr.py:214-283, one
    problem at a time -- the reference's own execution model, minus TensorFlow's dispatch) on the FIRST `instances` instances of the
    same workload, one core.  A stated baseline, not a target."""
    import trace_oracle
    from oracle import ilqr_ref
    x0 = np.asarray(x0[:instances].cpu() if hasattr(x0, "cpu") else x0[:instances], dtype=np.float32)
    u0 = np.asarray(u0[:instances].cpu() if hasattr(u0, "cpu") else u0[:instances], dtype=np.float32)
    its, t0 = 0, time.perf_counter()
    with np.errstate(all="ignore"):
        for b in range(len(x0)):
            o = ilqr_ref.ILQRRef(trace_oracle.make_env(kind, cfgs[b] if isinstance(cfgs, list) else cfgs, np.float32), dtype=np.float32,
                                 max_iterations=max_iterations)
            try:
                its += o.solve(x0[b].reshape(-1, 1), T, u_init=u0[b].reshape(T, -1, 1))[3] + 1
            except Exception:                                 # noqa: BLE001 -- (regularisation diverged on an fp32-unposable instance)
                its += max_iterations
    dt = time.perf_counter() - t0
    return {"value": its / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"{len(x0)} instances of {what}, fp32 numpy restatement of ilqr.py (oracle/ilqr_ref.py), one at a time, {dt:.1f} s"}


def roofline_hbm(alg_bytes, seconds, traffic):
    gbs = alg_bytes / seconds / 1e9
    return {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
            "algorithmic_bytes": alg_bytes, "traffic": traffic}


def ilqr_api_rate(n, m, T, B, reps=10):
    """Secondary number (not `value`), first-class since round 3: BASELINE.json's headline shape driven through
    tfmpc.solvers.ilqr.iLQR.solve on the LQ env (whole iteration loops in one launch, matrix-core kernel).
    Problems: the reference's make_lqr distribution (tests/problems.py:make_lqr_batch_spd: C by make_spd_matrix's formula,
    eigenvalues ~1e-3 .. n + m) with ONE change: F is scaled by 0.25 (spectral radius ~1.3 instead of ~5).  iLQR's start is
    an OPEN-LOOP rollout of the given actions (ilqr.py:53-82), and at rho ~ 5 an open-loop rollout amplifies a rounding
    error 5^50 times over T = 50 -- it overflows fp32 in any implementation, also from the optimal actions.  Start
    actions: the LQR-optimal open-loop actions of each instance (LQR.solve on the same problem) perturbed by 5 % of
    their size -- a warm start near the optimum, as an MPC re-solve gives -- so every instance needs the full Newton
    step and the line search (mean ~2 iterations).  Iterations counted = reference loop index + 1.
    Roofline: algorithmic flop = backward passes x T x 45.0 kflop + rollouts x T x 2.2 kflop with
    backward passes = iterations and rollouts >= iterations - 1 (a converged instance ends on a backward pass)."""
    import workloads
    from tfmpc import _hip

    def run(solver, w, reps_):
        x0, u0 = w["x0"], w["u0"]
        out = solver.solve_device(x0, T, u_init=u0)
        if reps_ > 1:                                # (a few untimed launches: the first ones after host-side problem generation find the GPU at
            for _ in range(4):                       # idle clocks -- round 6: one warm-up launch read 3.98 ms where forty read 3.45)
                out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps_):
            out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) / reps_

    def api_line(w, reuse=True):
        out, dt = run(workloads.solver_of(w), w, reps)
        its = float((out["iterations"].double() + 1).sum())
        flop = its * T * 45.0e3 + max(its - B, 0.0) * T * 2.2e3
        tf = flop / dt / 1e12
        # EXECUTED flop (round 6): with gain reuse only the first backward pass of an instance is a full one (45.0 kflop a step); every
        # later pass is the vector recursion -- F~^T V_x (2 x 24 x 16), k = -Q_uu^-1 Q_u (2 x 64), K^T Q_u (2 x 128), 40 adds = 1 192 flop
        ex = (B * T * 45.0e3 + max(its - B, 0.0) * T * (1192.0 + 2.2e3)) if reuse else flop
        return {"iterations_per_s": its / dt, "ms_per_batch": dt * 1e3, "mean_iterations": its / B,
                "flagged_instances": int((out["status"] != 0).sum()), "workload": w["text"], "workload_version": w["version"],
                "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_TFLOPS,
                             "algorithmic_flop": flop, "executed_flop": ex, "frac_executed": ex / dt / 1e12 / PEAK_F32_TFLOPS}}

    # the workloads are defined in tests/workloads.py, the module the decision-trace parity tests draw them from
    # (tests/test_ilqr_lq_trace_gpu.py); `workload_version` says which definition a number belongs to
    w_warm = workloads.ilqr_api_warm(B, n, m, T)
    res = api_line(w_warm)
    if CPU_BASELINES:
        res["cpu_baseline"] = ilqr_cpu_baseline("lq", [workloads.instance_cfg(w_warm, b) for b in range(8)], w_warm["x0"], w_warm["u0"], T, 100, 8, w_warm["version"])
    res["roofline"]["traffic"] = pmc_traffic("ilqr_lq_mfma_kernel", None, float((res["mean_iterations"]) * B), file_tag="ilqr_api_pmc")
    res["roofline"]["traffic_note"] = "HBM bytes of the whole launch (PMC); the counters are scaled by iterations"
    res["executed"] = pmc_executed("ilqr_lq_mfma_kernel", file_tag="ilqr_api_pmc")
    res["roofline"]["executed"] = res["executed"]
    attach_profile(res, "ilqr_lq_mfma_kernel", file_tag="ilqr_api_pmc")
    res["roofline"]["kernel"] = "ilqr_lq_mfma_kernel<true, true> (profiles/r06_ilqr_api_kernel_stats.csv: tools/ilqr_api_once.py under rocprofv3)"
    res["gain_reuse"] = ("default since round 6: the env is time-invariant LQ and every pass runs at mu = 0, so K_t, V_xx(t), Q_uu(t) do not depend on the "
                         "trajectory -- from the second backward pass on the kernel keeps K_t and Q_uu^-1 of the first and runs the vector recursion for "
                         "k_t, V_x alone (ilqr_lq_mfma.hip, REUSE).  `frac` divides the ALGORITHMIC flop (every pass counted as the full 45.0 kflop a step, "
                         "as the reference executes it) by the time; `frac_executed` counts what the kernel executed.  TFMPC_ILQR_LQ_REUSE=0 is the full "
                         "pass in every iteration: `full_pass_every_iteration`")
    with _hip.option("TFMPC_ILQR_LQ_REUSE", "0"):
        full = api_line(w_warm, reuse=False)
    res["full_pass_every_iteration"] = {"ms_per_batch": full["ms_per_batch"], "frac": full["roofline"]["frac"], "mean_iterations": full["mean_iterations"]}
    # the same problems from zero actions: the first rollout runs open loop through an unstable system, several step sizes are tried
    cold = api_line(workloads.ilqr_api_cold(B, n, m, T))
    cold["roofline"].update(note="rollouts counted as iterations - B: a lower bound here (the line search backtracks)", traffic=None)
    cold["roofline"]["frac_executed"] = None          # (the rollout count is a bound, so the executed count would be one too)
    res["cold_start"] = cold
    # CONTROL LIMITS: box-QP at every backward step, regularisation loop in the kernel.  On the well-conditioned generator
    # (tests/problems.py:make_lqr_batch_fast, eigenvalues of C in [1, 2]; zero start actions) as in rounds 1-3: with
    # make_spd_matrix's spectrum and a +-0.5 box three quarters of the instances exhaust the regularisation attempts (the
    # reference would loop on them without bound, ilqr.py:238-270) -- a statement about those problems, not a benchmark.
    wl = workloads.control_limited(B, n, m, T)

    def limited(tag):
        out_, dt_ = run(workloads.solver_of(wl), wl, 1)
        its_ = float((out_["iterations"].double() + 1).sum())
        st, it = out_["status"], (out_["iterations"] + 1).float()
        return {"ms_per_batch": dt_ * 1e3, "solves_per_s": B / dt_, "iterations_per_s": its_ / dt_, "mean_iterations": its_ / B,
                "iterations_p50_p90_p99_max": [float(torch.quantile(it, q)) for q in (0.5, 0.9, 0.99, 1.0)],
                "instances_with_cholesky_retries": int(((st & 2) != 0).sum()),
                "instances_at_attempt_cap": int(((st & 16) != 0).sum()), "regularisation_search": tag}

    # ... and the same problems with a stable open loop (0.18 F): every instance is one fp32 can pose (tests/workloads.py)
    wl_run, wl = wl, workloads.control_limited_stable(B, n, m, T)
    stable = limited("linear probe 0, 1, 2, ... (ilqr.py:285-315; the default)")
    stable["workload"], stable["workload_version"] = wl["text"], wl["version"]
    stable["executed"] = pmc_executed("ilqr_lq_box_mfma_kernel<false, 0, true>", file_tag="box_stable")      # (the main launch -- the instantiation with helper teams --, not the sample probe)
    attach_profile(stable, "ilqr_lq_box_mfma_kernel<false, 0, true>", file_tag="box_stable")
    stable["helper_teams"] = ("8 teams x 5 helper blocks of the same launch roll out the step sizes of the longest instances' line searches side by side "
                              "(DESIGN.md 3.6; same bits: tests/test_ilqr_lq_box_mfma_gpu.py); TFMPC_BOX_HELPERS=off is the launch without them")
    with _hip.option("TFMPC_BOX_HELPERS", "off"):
        stable["without_helper_teams_ms"] = limited("")["ms_per_batch"]
    if CPU_BASELINES:
        stable["cpu_baseline"] = ilqr_cpu_baseline("lq", [workloads.instance_cfg(wl, b) for b in range(8)], wl["x0"], wl["u0"], T, 100, 8, wl["version"])
    # ... and with a box so wide (+-2) that the box-QP rarely clamps while the rollout's clip still bites: the one LQ line on which the line
    # search BACKTRACKS (an LQ problem's Newton step is exact, so the unbounded lines above accept the first step size every time);
    # rollouts per iteration measured from the decision trace of the same solve
    wl = workloads.control_limited_stable(B, n, m, T, bound=2.0)
    wide = limited("linear probe (default)")
    tr = workloads.solver_of(wl).solve_device(wl["x0"], wl["T"], u_init=wl["u0"], trace_rows=170)
    rows, ln = tr["trace"], tr["trace_len"]
    valid = torch.arange(rows.shape[1], device=rows.device)[None, :] < ln[:, None]
    searched = valid & (rows[..., 8] >= 0)
    wide["rollouts_per_iteration_measured"] = float((rows[..., 5] + 1)[searched].sum()) / max(float((tr["iterations"].double() + 1).sum()), 1.0)
    wide["passes_whose_line_search_backtracked"] = int((searched & (rows[..., 5] > 0)).sum())
    wide["workload"], wide["workload_version"] = wl["text"] + " -- box widened to +-2", wl["version"] + "+box2"
    del tr, rows
    stable["wide_box_variant"] = wide
    wl = wl_run
    res["control_limited"] = limited("linear probe 0, 1, 2, ... (ilqr.py:285-315; the default)")
    res["control_limited"]["stable_open_loop_variant"] = stable
    res["control_limited"]["note"] = ("8 701 instances of this batch start from an open-loop rollout of cost 1e12 .. 1e21 (0.25 F is unstable for them): "
                                      "their Cholesky retries and 100-iteration crawls are fp32 artefacts that the fp32 restatement reproduces and "
                                      "the fp64 one does not have (profiles/r04_box_family_oracle.json, r04_box_first_level.txt); they are most of this "
                                      "launch's time.  `stable_open_loop_variant` is the same workload without them.")
    res["control_limited"]["workload"] = wl["text"]
    res["control_limited"]["workload_version"] = wl["version"]
    res["control_limited"]["kernel"] = "ilqr_lq_box_mfma_kernel; round 1: wave kernel, 2.3 k solves/s at B=8192"
    with _hip.option("TFMPC_ILQR_RETRY", "bracket"):
        res["control_limited"]["bracket_search_variant"] = limited(
            "TFMPC_ILQR_RETRY=bracket: around the previous pass's level (another regularisation path on ~0.5 % of the instances, DESIGN.md 3.6)")
    return res


def torchenv_rate(B=1024, T=40):
    """Secondary number: an arbitrary differentiable env given as torch functions (`TorchEnv`; here the
    Navigation env of nav.config.json written in plain torch) -- derivatives by torch.func on the GPU, the backward pass in
    the HIP kernel, line-search rollouts as batched torch ops, the iteration loop driven from the host
    (tfmpc/solvers/ilqr.py:_solve_host_driven).  Iterations per second of one batched solve."""
    import problems
    from tfmpc.envs.torchenv import TorchEnv
    from tfmpc.solvers.ilqr import iLQR
    cfg = problems.NAV_CONFIG
    goal = torch.tensor(np.array(cfg["goal"], dtype=np.float32).reshape(-1), device="cuda")
    centers = torch.tensor(np.array(cfg["deceleration"]["center"], dtype=np.float32).reshape(-1, 2), device="cuda")
    decay = torch.tensor(np.array(cfg["deceleration"]["decay"], dtype=np.float32).reshape(-1), device="cuda")

    def transition(x, u):
        r = torch.linalg.norm(x[None, :] - centers, dim=1)
        lam = torch.prod(2.0 / (1.0 + torch.exp(-decay * r)) - 1.0)
        return x + lam * u

    def cost(x, u):
        return torch.sum((x - goal) ** 2)

    env = TorchEnv(transition, cost, lambda x: torch.sum((x - goal) ** 2), 2, 2,
                   low=np.array(cfg["low"], dtype=np.float32).reshape(-1, 1), high=np.array(cfg["high"], dtype=np.float32).reshape(-1, 1))
    solver = iLQR(env, max_iterations=10, compile_env=False)     # (the HOST-driven loop: what an env the translator cannot take still gets)
    rng = np.random.default_rng(4)
    x0 = rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
    u0 = solver.random_actions(T, B, seed=4)
    solver.solve_device(x0, T, u_init=u0)                        # warm-up: torch.func traces, allocator
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = solver.solve_device(x0, T, u_init=u0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    its = float((out["iterations"].double() + 1).sum())
    return {"iterations_per_s": its / dt, "ms_per_batch": dt * 1e3, "mean_iterations": its / B, "batch": B, "horizon": T,
            "workload": "Navigation (nav.config.json) as torch functions through TorchEnv: host-driven loop, <= 10 iterations",
            "note": "the rollout and derivative blocks of the loop are replayed as hipGraphs (iLQR(env, graphs=True)); "
                    "the built-in envs run fused"}


def mpc_rate(B=16384, T=20):
    """Secondary number: batched receding-horizon MPC episodes (agents/mpc.py:10-15, runners/__init__.py:14-43 of the
    reference, one episode per instance): T control steps, each = one fused iLQR re-solve over the remaining horizon + one env step with
    its noise model.  Wall time per control step, cold restart (the reference's: random actions every step) and warm start (the previous
    solution shifted by one step) -- tools/mpc_rates.py prints the full table."""
    import problems
    from tfmpc import agents, runners
    from tfmpc.envs.navigation import Navigation
    from tfmpc.envs.reservoir import Reservoir
    from tfmpc.solvers.ilqr import iLQR
    out = {"batch": B, "control_steps": T, "workload": "B episodes of T control steps, eager loop, one fused iLQR launch per control step"}
    for name, env, x0 in (("navigation", Navigation.load(problems.NAV_CONFIG), np.random.default_rng(1).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)),
                          ("res4", Reservoir.load(dict(problems.RES4_CONFIG)), np.tile(np.array(problems.RES4_X0, dtype=np.float32)[None], (B, 1, 1)))):
        for warm in (False, True):
            agent = agents.MPC(iLQR(env), T, warm_start=warm, seed=3)
            runner = runners.Runner(env, agent)
            env.seed(3)
            for _ in range(2):                                # (the second episode batch is the timed one)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with runner(x0, T) as r:
                    traj = r.run()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            out[f"{name}_{'warm' if warm else 'cold'}_start"] = {
                "ms_per_control_step": dt / T * 1e3, "ms_per_episode_batch": dt * 1e3,
                "mean_ilqr_iterations_per_resolve": float(np.mean([np.mean(i) for i in agent.iterations])),
                "mean_total_cost": float(np.mean(traj.total_cost))}
    return out


def deviceenv_rate(B=16384, T=50):
    """Secondary number (SURVEY.md 8f N2 at hot-path speed, round 5): BASELINE configs[3]'s Navigation env given as C++ DEVICE source
    (tests/deviceenv_sources.py) to tfmpc.envs.deviceenv.DeviceEnv -- compiled with hipcc when first used, derivatives by forward-mode dual
    numbers one direction per lane, the fused wave-per-instance solve kernel -- beside the same env on the built-in kernels."""
    import deviceenv_sources as sources
    import problems
    from tfmpc import _hip
    from tfmpc.envs.deviceenv import DeviceEnv
    from tfmpc.envs.navigation import Navigation
    from tfmpc.solvers.ilqr import iLQR
    cfg = problems.NAV_CONFIG
    rng = np.random.default_rng(4)
    x0 = torch.as_tensor(rng.uniform(0, 10, size=(B, 2, 1)).astype(np.float32), device="cuda")      # (resident before the timed regions)
    builtin = iLQR(Navigation.load(cfg))
    u0 = builtin.random_actions(T, B, seed=4)
    t0 = time.perf_counter()
    user = iLQR(DeviceEnv(sources.NAVIGATION, 2, 2, params=sources.navigation_params(cfg), low=np.array(cfg["low"]), high=np.array(cfg["high"])))
    user.env._library()
    ready = time.perf_counter() - t0

    def timed(solver, option=None, reps=5):
        # (three untimed solves, then `reps` back to back inside one event pair: one solve timed from the host right after a single warm-up
        # scattered by +- 8 % from run to run -- clock ramp and host launch time of a 8 ms launch)
        with _hip.option("TFMPC_ILQR_KERNEL", option):
            out = None
            for _ in range(3):
                out = solver.solve_device(x0, T, u_init=u0, workspace=None if out is None else out["workspace"])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                out = solver.solve_device(x0, T, u_init=u0, workspace=out["workspace"])
            e1.record()
            torch.cuda.synchronize()
            dt = e0.elapsed_time(e1) * 1e-3 / reps
        return dt, float((out["iterations"].double() + 1).sum())
    dt_u, its_u = timed(user)
    kernel = user.last_kernel
    user.env._library().force_wave_kernel(True)                 # the kernel every other shape of user env runs on
    try:
        dt_uw, _ = timed(user)
    finally:
        user.env._library().force_wave_kernel(False)
    dt_w, its_w = timed(builtin, "wave")
    dt_g, its_g = timed(builtin)
    # ... and the same env written as PLAIN TORCH FUNCTIONS (tests/torch_envs.py), translated to device source by TorchEnv.to_device_env()
    # (tfmpc/envs/fxsource.py, round 6): what a user of the reference has is Python methods
    from_python = {}
    try:
        import torch_envs
        t0 = time.perf_counter()
        py = iLQR(torch_envs.navigation(cfg, "cuda").to_device_env())
        py.env._library()
        ready_py = time.perf_counter() - t0
        dt_p, its_p = timed(py)
        from_python = {"iterations_per_s": its_p / dt_p, "ms_per_batch": dt_p * 1e3, "mean_iterations": its_p / B, "trace_translate_compile_s": ready_py,
                       "ratio_to_hand_written_device_source": dt_p / dt_u, "kernel": py.last_kernel,
                       "workload": "Navigation as three plain torch functions -> TorchEnv(...).to_device_env() (make_fx trace -> device templates)"}
        # ... and the reference's own res4 / hvac6 configs written in plain torch: costs the translator PROVES piecewise linear -> the costate form of
        # the fused kernel, every step size of a line search at once (B = 16 384, T = 100, <= 12 iterations; beside the built-in env on the same
        # generic wave kernel, TFMPC_ILQR_KERNEL=wave, and on its specialised sixteen-per-wave kernel)
        import workloads
        small = {}
        for name, make, cfg_ in (("res4", torch_envs.reservoir, problems.RES4_CONFIG), ("hvac6", torch_envs.hvac, problems.HVAC6_CONFIG)):
            ws_ = workloads.small_env(name)

            def timed_small(solver, option=None):
                with _hip.option("TFMPC_ILQR_KERNEL", option):
                    out = solver.solve_device(ws_["x0"], ws_["T"], u_init=ws_["u0"])
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        out = solver.solve_device(ws_["x0"], ws_["T"], u_init=ws_["u0"], workspace=out["workspace"])
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t0) / 3, float((out["iterations"].double() + 1).sum())
            env_py = make(dict(cfg_), "cuda").to_device_env()
            dt_s, its_s = timed_small(iLQR(env_py, max_iterations=12))
            dt_w, _ = timed_small(ws_["solver"], "wave")
            dt_d, _ = timed_small(ws_["solver"])
            small[name] = {"ms_per_batch": dt_s * 1e3, "iterations_per_s": its_s / dt_s, "cost_proved_piecewise_linear": bool(env_py.zero_cost_hessian),
                           "builtin_env_same_generic_wave_kernel_ms": dt_w * 1e3, "builtin_env_specialised_kernel_ms": dt_d * 1e3}
        from_python["piecewise_linear_envs_B16384_T100"] = small
        # ... and one of 17 <= n + m <= 32 (HVAC, n = 12 rooms): thirty-two lanes per instance, two instances per wave (B = 8 192, T = 100, <= 12 iterations)
        from tfmpc.envs.hvac import HVAC
        cfg12 = problems.hvac_config(12, seed=5)
        env12 = torch_envs.hvac(cfg12, "cuda").to_device_env()
        s12 = iLQR(env12, max_iterations=12)
        x12 = np.random.default_rng(3).uniform(10.0, 30.0, size=(8192, 12, 1)).astype(np.float32)
        u12 = iLQR(HVAC.load(cfg12)).random_actions(100, 8192, seed=2)
        out12 = s12.solve_device(x12, 100, u_init=u12)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out12 = s12.solve_device(x12, 100, u_init=u12, workspace=out12["workspace"])
        torch.cuda.synchronize()
        dt12 = (time.perf_counter() - t0) / 3
        from_python["hvac12_B8192_T100"] = {"ms_per_batch": dt12 * 1e3, "iterations_per_s": float((out12["iterations"].double() + 1).sum()) / dt12,
                                            "kernel": s12.last_kernel}
        # ... and an env that is none of the reference's: a torque-limited pendulum (n = 2, m = 1) as device source -- dense tiny envs (n + m <= 4) run
        # the lane-group kernel since round 6 (B = 16 384, T = 50, <= 30 iterations)
        import deviceenv_sources as more_sources
        pend = DeviceEnv(more_sources.PENDULUM, 2, 1, params=np.array([0.05, 9.81, 0.1, 0.0, 1.0, 0.1, 0.01], dtype=np.float32), low=-4.0, high=4.0)
        rp = np.random.default_rng(5)
        xp = np.stack([rp.uniform(-1.2, 1.2, size=16384), rp.uniform(-1.0, 1.0, size=16384)], axis=1).astype(np.float32)[..., None]
        up = np.zeros((16384, 50, 1, 1), dtype=np.float32)
        sp = iLQR(pend, max_iterations=30)
        outp = sp.solve_device(xp, 50, u_init=up)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            outp = sp.solve_device(xp, 50, u_init=up, workspace=outp["workspace"])
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / 5
        from_python["pendulum_device_source_B16384_T50"] = {"ms_per_batch": dtp * 1e3, "iterations_per_s": float((outp["iterations"].double() + 1).sum()) / dtp,
                                                            "kernel": sp.last_kernel}
    except Exception as exc:                                  # noqa: BLE001
        from_python = dict(from_python, error=repr(exc))
    return {"from_python_functions": from_python, "iterations_per_s": its_u / dt_u, "ms_per_batch": dt_u * 1e3, "mean_iterations": its_u / B, "batch": B, "horizon": T,
            "library_ready_s": ready, "kernel": kernel, "user_env_on_the_generic_wave_kernel_ms": dt_uw * 1e3,
            "same_env_builtin_generic_wave_kernel_ms": dt_w * 1e3, "same_env_builtin_lane_group_kernel_ms": dt_g * 1e3,
            "ratio_to_builtin_kernel": dt_u / dt_g,
            "workload": "Navigation (nav.config.json) as DeviceEnv source: transition / cost / final_cost as C++ device functions, derivatives by dual numbers",
            "note": "a 2 x 2 user env runs the lane-group kernel (16 lanes per instance, persistent groups) like the built-in 2-D envs; any other "
                    "shape the generic wave kernel, at the speed of a built-in env there"}


def other_config_rates():
    """Secondary numbers (not `value`): the other BASELINE.json configs, timed launches after a warm-up --
    cfg2 navlin LQR (B=4096), cfg4 Navigation iLQR (B=16384: single batch, AND one launch of 8 x 16384 instances through the persistent kernel), cfg5 HVAC / Reservoir
    iLQR (n=32, T=100, B=32768, <= 12 iterations; shared env: 16 instances per wave, coupling products on the matrix
    cores), the reference's own hvac6 / res4 configs (B=16384), configs[4]'s literal dims (n=32, m=16) as iLQR on the
    LQ env, and a dense LQR beyond the headline tile (n=32, m=16, B=8192).  Inputs as in SURVEY.md 8(d).  HBM-bound
    configs carry a `roofline` block: algorithmic bytes per iteration x iterations / time against 8 TB/s,
    `traffic` = PMC-measured HBM bytes of the same launch (the newest profiles/r*_pmc.json taken on the kernel's current sources)."""
    import problems
    import workloads
    from tfmpc.envs import make_lqr_linear_navigation
    from tfmpc.envs.hvac import HVAC
    from tfmpc.envs.lq import LQEnv
    from tfmpc.envs.navigation import Navigation
    from tfmpc.envs.reservoir import Reservoir
    from tfmpc.solvers.ilqr import iLQR
    from tfmpc.solvers.lqr import LQR

    def timed(fn, reps):
        out = fn(None)                               # allocates the workspace
        for _ in range(6 if reps >= 3 else 1):       # untimed launches: the first ones after host-side problem generation find the GPU idle
            out = fn(out["workspace"])               # (lower clocks; round 6: one warm-up read 1.49 ms where forty read 1.24 on the n = 32 LQR line)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn(out["workspace"])
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) / reps

    def ilqr_line(solver, x0, T, u0, reps, alg_bytes=None, pmc=None):
        out, dt = timed(lambda ws: solver.solve_device(x0, T, u_init=u0, workspace=ws), reps)
        its = float((out["iterations"].double() + 1).sum())
        line = {"ms_per_batch": dt * 1e3, "iterations_per_s": its / dt, "mean_iterations": its / x0.shape[0],
                "flagged_instances": int((out["status"] != 0).sum()), "batch": int(x0.shape[0]), "horizon": T,
                "launches_timed": reps}       # (back to back on one stream, one synchronize behind the last: the contract's K steps)
        if alg_bytes is not None:
            line["roofline"] = roofline_hbm(alg_bytes * its, dt, pmc_traffic(pmc[0], pmc[1], its, file_tag=pmc[2]) if pmc else None)
            line["roofline"]["algorithmic_bytes_per_iteration"] = alg_bytes
        if pmc:
            attach_profile(line, pmc[0], file_tag=pmc[2])
        return line

    res = {}
    w2 = workloads.cfg2()
    lqr, x0n_d = w2["lqr"], w2["x0"]                                                     # (resident before the timed region)
    _, dt = timed(lambda ws: lqr.solve_device(x0n_d, 50, workspace=ws), 200)     # (0.05 ms per launch: 200 launches behind one synchronize)
    # cfg2: n = m = 2, T = 50, F and C shared by the batch: read x0, goal 16 B, write x, u, c 253 floats = 1 012 B per solve
    res["cfg2_navlin_lqr"] = {"ms_per_batch": dt * 1e3, "solves_per_s": 4096 / dt, "batch": 4096, "horizon": 50, "launches_timed": 200,
                              "kernel": _hip_kernel_name(2, 2, 50),
                              "roofline": dict(roofline_hbm(1028 * 4096, dt, pmc_traffic("lqr_lane", None, 4096, file_tag="cfg2")), algorithmic_bytes_per_solve=1028,
                                               note="4 096 solves are 64 wavefronts of a lane-per-instance kernel on a 1 024-SIMD chip: the launch is one wave's "
                                                    "dependent T = 50 recursion (latency), not bandwidth")}
    attach_profile(res["cfg2_navlin_lqr"], "lqr_lane", file_tag="cfg2")
    # cfg4: n = m = 2, T = 50 -> read x, u 808 B + write x, u, c 1 012 B per iteration
    w4 = workloads.cfg4()
    solver, x0, u0, Bn = w4["solver"], w4["x0"], w4["u0"], 16384                       # (resident before the timed region)
    res["cfg4_navigation_ilqr"] = ilqr_line(solver, x0, 50, u0, 5, alg_bytes=1820, pmc=("ilqr_group_solve", None, "cfg4_pmc"))
    ex4 = pmc_executed("ilqr_group_solve", file_tag="cfg4_pmc")
    if ex4 is not None:
        # HBM is the wrong yardstick for this kernel (30 GB/s of 8 TB/s): it is bound by the issue and latency of its own instruction stream.
        # Issue side: the rate the same instruction stream would reach with the vector unit busy every cycle.
        r4 = res["cfg4_navigation_ilqr"]
        r4["roofline_issue_side"] = {"bound": "valu_issue", "achieved": r4["iterations_per_s"], "unit": "iterations/s",
                                     "peak": r4["iterations_per_s"] / max(ex4["valu_busy_frac"], 1e-9), "frac": ex4["valu_busy_frac"],
                                     "executed": ex4, "note": "frac = share of the SIMD cycles the vector unit is busy in the profiled launch "
                                                              "(single batch: 3 072 waves, a straggler's dependent iterations set the time)"}
    if CPU_BASELINES:
        res["cfg4_navigation_ilqr"]["cpu_baseline"] = ilqr_cpu_baseline("navigation", problems.NAV_CONFIG, x0, u0, 50, 100, 4, "Navigation (nav.config.json), T=50")
    res["cfg4_navigation_ilqr"]["note"] = ("one launch lasts as long as its slowest instance (median 8 iterations, p99 20, max 87: "
                                           "profiles/r02_cfg4_iteration_histogram.json); a larger launch keeps the chip full (below); round 3: "
                                           "closed-form two-variable box-QP, hardware sqrt / exp2 / rcp in the env")
    # Sustained rate: ONE launch of 8 x 16 384 instances.  The group kernel is persistent since round 4 (ilqr_lane.hip: the grid is
    # what the chip holds at once, a group whose instance has finished takes the next one from an atomic queue), so the rate that
    # round 3 needed eight batches in flight on eight host streams (and a throw-away stream pool) for comes out of a single launch.
    w8 = workloads.cfg4(batches=8)
    x8, u8 = w8["x0"], w8["u0"]
    big = ilqr_line(solver, x8, 50, u8, 3, alg_bytes=1820, pmc=("ilqr_group_solve", None, "cfg4_one_launch"))
    big["executed"] = pmc_executed("ilqr_group_solve", file_tag="cfg4_one_launch")
    res["cfg4_navigation_ilqr"]["one_launch_of_8x16384_instances"] = big
    res["cfg4_navigation_ilqr"]["one_launch_of_8x16384_instances"]["note"] = (
        "persistent lane groups + instance queue; round 3: 8 streams x 16 384 = 51.1 M it/s (profiles/r03_cfg4_sustained.json)")
    del x8, u8, w8
    # cfg5: n = m = 32, T = 100 -> read x, u 25.7 KB + write x, u, c 26.1 KB = 51.8 KB per iteration
    for kind, kernel_tag in (("hvac", "ilqr_adjoint_mfma_kernel<3, 2"), ("reservoir", "ilqr_adjoint_mfma_kernel<100, 2")):     # two-tile instantiations (Reservoir: the chain form, tag 100)
        n, T, B = 32, 100, 32768
        w5 = workloads.cfg5(kind, B, n, T)            # the same problems tests/test_ilqr_teacher_forced_gpu.py holds against the restatement
        env, x0, u0c = w5["env"], w5["x0"], w5["u0"]
        solver = iLQR(env, max_iterations=12)
        line = ilqr_line(solver, x0, T, u0c, 8, alg_bytes=4 * (2 * (T + 1) * n + 2 * T * n + (T + 1)), pmc=(kernel_tag, None, f"cfg5_{kind}"))
        # The flop side (round 4): rollouts per iteration MEASURED from the decision trace of the same solve (a traced launch returns
        # the same bits): a pass of the reference's line search makes alpha_index + 1 rollouts (ilqr.py:322-353).  Algorithmic flop
        # per iteration, SURVEY.md 8(d) dense count: costate sweep T x 4 n^2 + rollouts x T x (4 n^2 + env), env = 20 n flop per step
        # (element-wise transition + cost); against the fp32 peak.  Which side binds is whichever fraction is larger.
        tr = solver.solve_device(x0, T, u_init=u0c, trace_rows=int(solver.max_iterations) + int(solver.max_attempts) + 1)
        rows, ln = tr["trace"], tr["trace_len"]
        valid = torch.arange(rows.shape[1], device=rows.device)[None, :] < ln[:, None]
        searched = valid & (rows[..., 8] >= 0)
        rollouts = float((rows[..., 5] + 1)[searched].sum())
        passes = float(valid.sum())
        its_c = line["iterations_per_s"] * line["ms_per_batch"] * 1e-3
        flop = T * 4 * n * n * passes + rollouts * T * (4 * n * n + 20 * n)
        tfl = flop / (line["ms_per_batch"] * 1e-3) / 1e12
        line["rollouts_per_iteration_measured"] = rollouts / max(its_c, 1.0)
        line["backward_passes_per_iteration_measured"] = passes / max(its_c, 1.0)
        # NOT a utilisation figure: SURVEY's dense count (4 n^2 per step and product) divided by time; the kernel evaluates the env's
        # Jacobians in closed form and, on Reservoir, replaces the products by row moves -- see `executed` for what the units did
        line["algorithmic_flop_rate"] = {"bound": "mfma", "achieved": tfl, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": tfl / PEAK_F32_TFLOPS,
                                      "algorithmic_flop": flop, "algorithmic_flop_per_iteration": flop / max(its_c, 1.0),
                                      "flop_per_algorithmic_byte": flop / max(its_c, 1.0) / (4 * (2 * (T + 1) * n + 2 * T * n + (T + 1))),
                                      "note": "dense count of SURVEY.md 8(d) with the measured rollouts; the ridge is 157.3 TF / 8 TB/s = 19.7 flop/B"}
        del tr, rows
        line["workload_version"] = w5["version"]
        line["executed"] = pmc_executed(kernel_tag, file_tag=f"cfg5_{kind}") if (line.get("profile") or {}).get("reproduces_within_5_percent", True) else None
        # 43 - 57 algorithmic flop per algorithmic byte (ridge 19.7): this line sits on the COMPUTE side, and the units that compute here are the
        # vector ALUs (closed-form Jacobians, row moves; the matrix cores carry only HVAC's coupling products) -- so the yardstick the line LEADS
        # with is the share of SIMD cycles the vector unit is busy in the profiled launch; the HBM block comes second
        line["roofline_hbm"] = line["roofline"]
        if line["executed"]:
            line["roofline"] = {"bound": "valu", "achieved": line["executed"]["valu_busy_frac"], "peak": 1.0, "unit": "share of SIMD cycles with the vector unit busy (PMC)",
                                "frac": line["executed"]["valu_busy_frac"], "traffic": line["roofline_hbm"].get("traffic"),
                                "mfma_busy_frac": line["executed"].get("mfma_busy_frac"), "source": line["executed"].get("source")}
        if CPU_BASELINES:
            line["cpu_baseline"] = ilqr_cpu_baseline(kind, w5["cfg"], x0, u0c, T, 12, 3, w5["version"])
        res[f"cfg5_{kind}_ilqr_n32"] = line
    # the reference's own env configs (hvac6.config.json n = 6, res4.config.json n = 4) at a large batch
    for name, env, x0r, kernel_tag in (("hvac6", HVAC.load(dict(problems.HVAC6_CONFIG)), problems.HVAC6_X0, "ilqr_adjoint_mfma_kernel<3, 1"),
                                       ("res4", Reservoir.load(dict(problems.RES4_CONFIG)), problems.RES4_X0, "ilqr_adjoint_mfma_kernel<4, 1")):
        B, T = 16384, 100
        ws_ = workloads.small_env(name, B, T)
        n, x0, solver = ws_["n"], ws_["x0"], ws_["solver"]                              # (resident before the timed region)
        # PMC: profiles/r0x_small_env_pmc.json (tools/small_env_once.py, 16 384 instances x 12 iterations)
        res[f"{name}_reference_config_ilqr"] = ilqr_line(solver, x0, T, ws_["u0"], 10,      # (ten launches back to back: one sync's latency over 20 ms, not over 4)
                                                         alg_bytes=4 * (2 * (T + 1) * n + 2 * T * n + (T + 1)),
                                                         pmc=(kernel_tag, None, name))
    # configs[4] at its literal dims (n = 32, m = 16, T = 100, B = 32 768) as iLQR on the generalised LQ env
    n, m, T, B = 32, 16, 100, 32768
    import workloads
    wl = workloads.literal_dims(B, n, m, T)
    solver = workloads.solver_of(wl)
    x0d = wl["x0"]
    line = ilqr_line(solver, x0d, T, wl["u0"], 2)
    line["workload"], line["workload_version"] = wl["text"], wl["version"]
    # algorithmic flop (SURVEY.md 8d formulas at n = 32, m = 16): backward 352.6 kflop per step, rollout 8.8 kflop per step
    its = line["iterations_per_s"] * line["ms_per_batch"] * 1e-3
    bw_step = 4 * n ** 3 + 4 * m * n * n + 2 * m * m * n + 2 * n * n + 2 * m * n + m ** 3 / 3.0 + 2 * m * m * (n + 1) \
        + 6 * n * n * m + 2 * n * m * m + 6 * n * m + 2 * m * m + 2 * m
    fw_step = 2 * m * n + 2 * n * (n + m) + 2 * (n + m) ** 2 + 2 * (n + m)
    flop = its * T * bw_step + max(its - B, 0.0) * T * fw_step
    tf = flop / (line["ms_per_batch"] * 1e-3) / 1e12
    line["roofline"] = {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_TFLOPS,
                        "algorithmic_flop": flop,
                        # PMC: profiles/r04_large_tile_pmc.json, taken at 8 192 instances x 2.02 iterations (tools/large_tile_once.py)
                        "traffic": pmc_traffic("ilqr_lq_mfma32_kernel", None, its, file_tag="literal_dims")}
    attach_profile(line, "ilqr_lq_mfma32_kernel", file_tag="literal_dims")
    line["kernel"] = "ilqr_lq_mfma32_kernel (2 x 2 tiles of bf16x3, trajectories in HBM); round-2 start: wave kernel, 1 123 ms"
    res["cfg5_literal_dims_ilqr_lq_n32_m16"] = line
    del solver, x0d
    w32 = workloads.lqr32()
    big, x0d = w32["lqr"], w32["x0"]
    _, dt = timed(lambda ws: big.solve_device(x0d, 50, workspace=ws), 20)
    tf = lqr_flops_per_solve(32, 16, 50) * 8192 / dt / 1e12
    res["lqr_n32_m16"] = {"ms_per_batch": dt * 1e3, "solves_per_s": 8192 / dt, "batch": 8192, "horizon": 50,
                          "kernel": _hip_kernel_name(32, 16, 50),
                          "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                                       "frac": tf / PEAK_F32_TFLOPS, "algorithmic_flop_per_solve": lqr_flops_per_solve(32, 16, 50),
                                       "traffic": pmc_traffic("lqr_mfma32x16_kernel", None, 8192, file_tag="lqr32")}}
    attach_profile(res["lqr_n32_m16"], "lqr_mfma32x16_kernel", file_tag="lqr32")
    return res


def _hip_kernel_name(n, m, T):
    from tfmpc import _hip
    return _hip.require_gpu().tfmpc_lqr_kernel_name(n, m, T).decode()


def numpy_single_instance_rate(n, m, T, instances=24):
    """The reference's execution model -- one problem at a time, fp32, its operation order -- as
    the oracle's numpy restatement looped over a few instances on ONE core.
    Optimistic for the reference: no TensorFlow op-dispatch or graph-tracing overhead."""
    import problems
    from oracle import lqr_ref
    F, f, C, c, x0 = problems.make_lqr_batch_spd(instances, n, m, seed=5)
    t0 = time.perf_counter()
    for b in range(instances):
        lqr_ref.solve(F[b], f[b], C[b], c[b], x0[b], T, dtype=np.float32)
    dt = time.perf_counter() - t0
    return {"value": instances / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"{instances} instances, numpy fp32 restatement of lqr.py, one at a time"}


def cpu_baseline(n, m, T, target_seconds=12.0):
    """The oracle's C port of the reference equations, OpenMP over instances on all
    host cores, on a bounded sample of the same workload."""
    from oracle import c_oracle
    import problems

    cores = os.cpu_count() or 1
    threads = max(1, min(cores, c_oracle.max_threads()))
    calib = 64 * threads
    F, f, C, c, x0 = problems.make_lqr_batch_spd(calib, n, m, seed=99)
    c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float32, nthreads=threads)       # warm up threads
    t0 = time.perf_counter()
    c_oracle.lqr_solve(F, f, C, c, x0, T, dtype=np.float32, nthreads=threads)
    rate = calib / (time.perf_counter() - t0)
    sample = int(min(BATCH, max(calib, rate * target_seconds)))
    reps = -(-sample // calib)
    Fs, fs, Cs, cs, xs = (np.concatenate([a] * reps)[:sample] for a in (F, f, C, c, x0))
    t0 = time.perf_counter()
    c_oracle.lqr_solve(Fs, fs, Cs, cs, xs, T, dtype=np.float32, nthreads=threads)
    dt = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "iterations/s", "cores": threads, "kind": "port",
            "sample": f"{sample} of {BATCH} instances of the same workload (n={n}, m={m}, T={T}, fp32), "
                      f"C restatement of lqr.py, OpenMP x{threads}, {dt:.1f} s",
            "host_cpus": cores}


def dry_run_cpu(args):
    """Same rank bookkeeping as the real run, gloo instead of RCCL, sleep instead of kernels.
    The printed line is marked invalid on purpose: it measures nothing."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from tfmpc.parallel import gather_trajectories
    from tfmpc.parallel import shard_bounds
    # strong: a FIXED global batch split N ways; weak: a fixed batch per rank.  Uneven block splits on purpose
    # (strong: 129 over 2 ranks = 65 + 64; weak: 64 per rank + 1: the first rank gets one instance more)
    n, m, T = 4, 2, 5
    B_total = 129 if args.scaling == "strong" else 64 * world + 1
    lo_, hi_ = shard_bounds(B_total, world, rank)
    B = hi_ - lo_

    def fence():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        time.sleep(0.001)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))         # rank 1 is slower: MAX over ranks must pick it up
    fence()
    elapsed = time.perf_counter() - t0
    states = torch.full((B, T + 1, n, 1), float(rank))
    # total= : the shard sizes follow from the block split, so the gather is the ONLY collective of the data path
    gathered = gather_trajectories(states, torch.zeros(B, T, m, 1), torch.zeros(B, T + 1), total=B_total) if world > 1 else None
    total = torch.tensor([float(B)])
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        dist.all_reduce(total)
    if rank == 0:
        line = {"metric": "DRY RUN (no GPU work)", "value": float(total.item()) * args.steps / elapsed, "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "valid": False,
                "scaling": args.scaling, "total_instances": int(total.item())}
        if gathered is not None:
            line["gathered_states_shape"] = list(gathered[0].shape)
            line["gathered_rank_of_last_instance"] = float(gathered[0][-1, 0, 0, 0])
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a torch.distributed.run child
    process (never a re-exec: this process has not touched the GPU and never will) and hand back its exit code."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def spawn_extras(args):
    """Start the child that measures `extra`; it blocks on stdin until collect_extras() releases it."""
    cmd = [sys.executable, os.path.abspath(__file__), "--extras-only", "--batch", str(args.batch)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    return subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def collect_extras(child, timeout=900):
    """Release the child, wait for its JSON object; any failure becomes an `error` entry, never an exception."""
    try:
        out, err = child.communicate("go\n", timeout=timeout)
        if child.returncode != 0:
            return {"error": f"the extras process exited with {child.returncode}", "stderr_tail": err[-600:]}
        return json.loads(out.strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        return {"error": f"the extras process did not finish within {timeout} s"}
    except Exception as exc:                                  # noqa: BLE001 -- secondary numbers must never cost the headline line
        return {"error": repr(exc)}


def summarise_extras(extra):
    """One short dict of the secondary numbers (ms per batch, M iterations/s, roofline fraction), keyed by workload."""
    def get(d, *keys):
        for k in keys:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    r3 = lambda v: None if v is None else float(f"{v:.3g}")
    oc = extra.get("other_configs", {}) if isinstance(extra, dict) else {}
    api = extra.get("ilqr_api", {}) if isinstance(extra, dict) else {}
    out = {"ilqr_api_warm": [r3(get(api, "ms_per_batch")), r3(get(api, "roofline", "frac")), r3(get(api, "roofline", "frac_executed"))],
           "ilqr_api_warm_full_pass_every_iteration": [r3(get(api, "full_pass_every_iteration", "ms_per_batch")), r3(get(api, "full_pass_every_iteration", "frac"))],
           "ilqr_api_cold": [r3(get(api, "cold_start", "ms_per_batch")), r3(get(api, "cold_start", "roofline", "frac"))],
           "control_limited_ms": r3(get(api, "control_limited", "ms_per_batch")),
           "control_limited_stable_ms": r3(get(api, "control_limited", "stable_open_loop_variant", "ms_per_batch")),
           "cfg4_single_batch_ms": r3(get(oc, "cfg4_navigation_ilqr", "ms_per_batch")),
           "cfg4_one_launch_8x16384_Mit_s": r3((get(oc, "cfg4_navigation_ilqr", "one_launch_of_8x16384_instances", "iterations_per_s") or 0) / 1e6),
           "bf16_sweep": get(extra, "bf16_storage_sweep") if isinstance(extra, dict) and "error" not in (extra.get("bf16_storage_sweep") or {}) else None,
           "torchenv_kit_s": r3((get(extra, "torchenv_generic_env", "iterations_per_s") or 0) / 1e3),
           "deviceenv_Mit_s": r3((get(extra, "deviceenv_user_env", "iterations_per_s") or 0) / 1e6),
           "mpc_ms_per_control_step_nav_cold_warm_res4_cold_warm": [r3(get(extra, "mpc", k, "ms_per_control_step")) for k in
                                                                    ("navigation_cold_start", "navigation_warm_start", "res4_cold_start", "res4_warm_start")],
           "deviceenv_from_python_Mit_s": r3((get(extra, "deviceenv_user_env", "from_python_functions", "iterations_per_s") or 0) / 1e6),
           "res4_hvac6_from_python_ms": [r3(get(extra, "deviceenv_user_env", "from_python_functions", "piecewise_linear_envs_B16384_T100", k, "ms_per_batch")) for k in ("res4", "hvac6")],
           "hvac12_from_python_ms": r3(get(extra, "deviceenv_user_env", "from_python_functions", "hvac12_B8192_T100", "ms_per_batch")),
           "pendulum_device_source_Mit_s": r3((get(extra, "deviceenv_user_env", "from_python_functions", "pendulum_device_source_B16384_T50", "iterations_per_s") or 0) / 1e6),
           "format": "[ms per batch, roofline frac (, algorithmic flop rate / fp32 peak; ilqr_api_warm: executed flop rate / fp32 peak)]"}
    for key, short in (("cfg5_hvac_ilqr_n32", "cfg5_hvac"), ("cfg5_reservoir_ilqr_n32", "cfg5_reservoir"), ("hvac6_reference_config_ilqr", "hvac6"),
                       ("res4_reference_config_ilqr", "res4"), ("cfg5_literal_dims_ilqr_lq_n32_m16", "literal_dims"), ("lqr_n32_m16", "lqr_n32_m16")):
        v = [r3(get(oc, key, "ms_per_batch")), r3(get(oc, key, "roofline", "frac"))]
        if get(oc, key, "algorithmic_flop_rate", "frac") is not None:
            v.append(r3(get(oc, key, "algorithmic_flop_rate", "frac")))
        out[short] = v
    if isinstance(out.get("bf16_sweep"), dict):
        out["bf16_sweep"] = {k.replace("hvac_", "").replace("_state_rel_err_vs_fp64", "").replace("_total_cost_rel_diff_bf16_vs_fp32", "_cost"): r3(v)
                             for k, v in out["bf16_sweep"].items() if isinstance(v, float)}
    return out


def extras_only(args):
    """Child side: secondary numbers of a single-GPU run (the iLQR-API line, the other BASELINE configs, the numpy rate)."""
    import torch                                              # (imported while waiting: importing does not touch the GPU)
    if not sys.stdin.readline():
        return 1                                              # the parent went away before the headline was measured
    n, m, T, B = N_STATE, N_ACTION, HORIZON, args.batch
    global CPU_BASELINES
    CPU_BASELINES = not args.no_cpu_baseline
    extra = {}
    try:
        extra["ilqr_api"] = ilqr_api_rate(n, m, T, B)
        extra["other_configs"] = other_config_rates()
        try:
            # BASELINE configs[4] "fp32 vs bf16 tolerance sweep": three numbers of the table tests/test_bf16_storage_sweep_gpu.py
            # asserts on (full table: profiles/r04_bf16_storage_sweep.json), measured here on a smaller batch
            import bf16_sweep
            extra["bf16_storage_sweep"] = dict(bf16_sweep.headline(bf16_sweep.sweep(B=256, n_oracle=2, kinds=("hvac",))),
                                               note="iLQR on HVAC n=m=32 T=100, 16-bit trajectory containers vs fp32 ones, against the fp64 "
                                                    "restatement (2 instances) / over 256 instances")
        except Exception as exc:                              # noqa: BLE001
            extra["bf16_storage_sweep"] = {"error": repr(exc)}
        try:
            extra["torchenv_generic_env"] = torchenv_rate()
        except Exception as exc:                              # noqa: BLE001
            extra["torchenv_generic_env"] = {"error": repr(exc)}
        try:
            extra["mpc"] = mpc_rate()
        except Exception as exc:                              # noqa: BLE001
            extra["mpc"] = {"error": repr(exc)}
        try:
            extra["deviceenv_user_env"] = deviceenv_rate()
        except Exception as exc:                              # noqa: BLE001
            extra["deviceenv_user_env"] = {"error": repr(exc)}
    except Exception as exc:                                  # noqa: BLE001
        extra["other_configs_error"] = repr(exc)
    if not args.no_cpu_baseline:
        extra["cpu_numpy_single_instance"] = numpy_single_instance_rate(n, m, T)
    print(json.dumps(extra), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH,
                    help="instances: the GLOBAL batch with --scaling strong, per GPU with --scaling weak (default = BASELINE config)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default; BASELINE configs[2]: batch=65 536 sharded over 1->8 GPUs): --batch is the global batch, "
                         "block-sharded over the ranks; weak: --batch instances on every GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary iLQR-API measurement (profiling runs)")
    ap.add_argument("--extras-only", action="store_true",
                    help="internal: the child process of a single-GPU run that measures the secondary numbers (`extra`); "
                         "it waits for a line on stdin before it touches the GPU and prints one JSON object")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="TEST ONLY: no GPU work; a fixed-sleep stand-in step drives the multi-rank control flow "
                         "(barriers, MAX over ranks, the final gather) over gloo so it can be tested without GPUs")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus is not None and args.gpus > 1:
            raise SystemExit(self_launch(args))         # before any GPU call in this process
    elif args.gpus is not None and args.gpus != int(os.environ["WORLD_SIZE"]):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    if args.dry_run_cpu:
        return dry_run_cpu(args)
    if args.extras_only:
        return extras_only(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under a launcher (torch.distributed.run: RANK / WORLD_SIZE set) the run goes through RCCL even with ONE rank, so
    # that the multi-GPU path -- process group, barriers, the gather -- can be rehearsed on a one-GPU box
    use_dist = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # The secondary numbers (`extra`) run in a CHILD process, started here -- before this process touches the GPU (no
    # exec after GPU initialisation) -- and idle until the headline measurement is over: whatever happens in one of the
    # secondary kernels, the headline line is printed.
    extras_child = spawn_extras(args) if (world == 1 and not use_dist and not args.no_extra) else None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from tfmpc import _hip
    from tfmpc.parallel import gather_buffers, gather_results
    from tfmpc.solvers.lqr import LQR
    import problems

    from tfmpc.parallel import shard_bounds
    n, m, T = N_STATE, N_ACTION, HORIZON
    # each rank owns its own contiguous shard of the global batch: strong = the block split of --batch instances over the
    # ranks (sizes differ by at most one), weak = --batch instances on every rank.  The shard is drawn on its rank
    # (seed 1234 + rank); at N = 1 both modes are the same 65 536 problems.
    B_global = args.batch if args.scaling == "strong" else args.batch * world
    lo_, hi_ = shard_bounds(B_global, world, rank)
    B = hi_ - lo_
    if B_global < world:               # (every rank sees this: nobody is left waiting in a collective)
        raise SystemExit(f"bench.py: --batch {args.batch} leaves a rank of {world} without an instance")
    F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=1234 + rank)
    lqr = LQR(F, f, C, c)
    x0_dev = lqr._prep_x0(x0)
    lib = _hip.require_gpu()
    kernel = lib.tfmpc_lqr_kernel_name(n, m, T).decode()
    ws = None

    def step():
        nonlocal ws
        out = lqr.solve_device(x0_dev, T, workspace=ws)
        ws = out["workspace"]
        return out

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the same steps with the sweep's products on the f32 matrix-core instruction (TFMPC_LQR_MFMA=f32): the strict
    # variant of the kernel, reported beside the default (bf16x3) -- not part of `value`.  Measured BEFORE the warm-up
    # and the timed steps: a process's first ~40 launches run 5-8 % slower than its steady state (1.92 -> 1.78 ms on one
    # box, tools/probes/stream_effect_headline.py), so what has to run anyway runs first.
    f32_ms, f32_launches = None, 0
    lqr_mfma_override = _hip.get_option("TFMPC_LQR_MFMA")     # an A/B run may have exported a variant: then the headline IS that variant
    if world == 1 and not use_dist and lqr_mfma_override is None:
        with _hip.option("TFMPC_LQR_MFMA", "f32"):
            for _ in range(3):
                step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(max(1, args.steps // 2)):
                step()
            e1.record()
            torch.cuda.synchronize()
        f32_ms = e0.elapsed_time(e1) / max(1, args.steps // 2)
        f32_launches = 3 + max(1, args.steps // 2)

    for _ in range(args.warmup):
        out = step()
    # the receive buffers of the final gather are allocated up front (rank 0), where running out of memory fails the job
    # at start-up; the gather itself is then ONE collective and nothing else (sizes follow from the block split)
    if use_dist:
        from tfmpc.parallel import check_shard_sizes
        check_shard_sizes(B, B_global)           # start-up agreement on the shard sizes (fails on every rank or none)
    recv = gather_buffers(out["states"], out["actions"], out["costs"], total=B_global, status=out["status"]) if use_dist else None
    fence()
    # ONE pair of HIP events around the K launches, on the stream the kernel is launched on: their difference / K is the
    # kernel's average duration INCLUDING the gap to the next launch.  (Rounds 1-4 bracketed every step with its own pair:
    # 2 K extra packets in the queue, which at the 8-GPU shard size -- 0.24 ms per launch -- showed up as 0.03 ms per step.)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = step()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps if args.steps else float("nan")

    status_bad = int((out["status"] != 0).sum())
    # the one collective of the path: gather the result trajectories on rank 0 (outside
    # the timed region: it happens once per job, not per step).  gather_trajectories fails on ALL ranks or none.
    gathered, gather_error, gather_ms = None, None, None
    if use_dist:
        fence()
        g0 = time.perf_counter()
        try:
            # (the per-instance status rides in the same packed buffer, bit-cast: still ONE collective)
            gathered = gather_results(out["states"], out["actions"], out["costs"], status=out["status"], total=B_global, recv=recv)
        except RuntimeError as exc:
            gather_error = repr(exc)
        fence()
        gather_ms = (time.perf_counter() - g0) * 1e3

    if use_dist:
        tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kmax = torch.tensor([kernel_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        kernel_ms = float(kmax.item())

    if rank == 0:
        total_solves = B_global * args.steps
        value = total_solves / elapsed
        # roofline of the dominant kernel on the slowest rank's clock: rank 0 holds a largest shard of the block split
        traffic, traffic_source = measured_traffic(kernel, B)
        flops = lqr_flops_per_solve(n, m, T) * B
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        line = {
            "metric": "iLQR iterations/sec (batch x horizon) at n=16,m=8,T=50",
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"random LQR n=16 m=8 horizon=50, global batch={B_global} block-sharded over {world} GPU(s) "
                                    "(BASELINE configs[2]); " if args.scaling == "strong" else
                                    f"random LQR n=16 m=8 horizon=50, batch={B} per GPU (BASELINE configs[2] on every GPU); ")
                                   + "one LQR solve = one iLQR iteration",
                       "generator": "tests/problems.py:make_lqr_batch_spd = the reference's make_lqr (tfmpc/envs/__init__.py:9-18) "
                                    "vectorised: F, f, c ~ N(0,1), C by sklearn make_spd_matrix's formula (eigenvalues ~1e-3 .. n+m)",
                       "state_dim": n, "action_dim": m, "horizon": T, "batch_per_gpu": B,
                       "global_batch": B_global, "parallelism": f"batch-sharded x{world}, one gather at the end",
                       "kernel": kernel},
            "timestep_iterations_per_s": value * T,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_TFLOPS,
                         "peak_note": "fp32 peak (vector == f32-MFMA dense peak); the default kernel's products run as bf16x3 on the bf16 matrix "
                                      "pipe (dense peak ~2.5 PFLOP/s), so this is the fp32-equivalent rate against the fp32 yardstick -- "
                                      "`strict_f32_variant` is the like-for-like figure",
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_unit": "HBM bytes per launch (PMC, profiles/)",
                         "traffic_note": "algorithmic bytes + the gain round trip: K_t, k_t (27.2 KB per solve) are produced "
                                         "backwards and consumed forwards, written once and read once = 3.57 GB per launch "
                                         "(DESIGN.md section 3.1)",
                         "kernel_ms": kernel_ms, "algorithmic_flop_per_iteration": lqr_flops_per_solve(n, m, T),
                         "algorithmic_bytes_per_iteration": lqr_bytes_per_solve(n, m, T),
                         "hbm_frac_at_algorithmic_bytes": lqr_bytes_per_solve(n, m, T) * B / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
            "status_flagged_instances": status_bad,
            # launches of this process before the timed region: the declared warm-up + the strict-variant pass above
            "untimed_launches_before_timing": args.warmup + f32_launches,
            "lqr_mfma_option": lqr_mfma_override or "default (bf16x3)",
        }
        if f32_ms is not None:
            flop_s = flops / (f32_ms * 1e-3) / 1e12
            line["strict_f32_variant"] = {"option": "TFMPC_LQR_MFMA=f32 (v_mfma_f32_16x16x4_f32 instead of bf16x3)",
                                          "kernel_ms": f32_ms, "achieved": flop_s, "frac": flop_s / PEAK_F32_TFLOPS}
            # inside `roofline` too, so that a reader who keeps only that block keeps the like-for-like figure
            line["roofline"]["frac_strict_f32"] = flop_s / PEAK_F32_TFLOPS
            line["roofline"]["kernel_ms_strict_f32"] = f32_ms
        if gathered is not None:
            line["gathered_states_shape"] = list(gathered[0].shape)
        if gather_ms is not None:
            from tfmpc.parallel import gather_bytes_per_rank
            per_rank = gather_bytes_per_rank(out["states"], out["actions"], out["costs"], status=out["status"])
            line["gather"] = {"collective": "ONE dist.gather (RCCL) of the packed trajectories onto rank 0, after the timed steps; "
                                            "shard sizes follow from the block split (no size exchange), receive buffers "
                                            "allocated at start-up",
                              "backend": dist.get_backend(), "world_size": world,
                              "ms": gather_ms, "bytes_per_rank": per_rank, "bytes_total": per_rank * world,
                              "GB_per_s_into_rank0": per_rank * (world - 1) / (gather_ms * 1e-3) / 1e9,
                              "checked": bool(gathered is not None and torch.equal(gathered[0][:B], out["states"])
                                              and torch.equal(gathered[4][:B], out["status"]))
                              if world == 1 else None}
        if gather_error is not None:
            line["gather_error"] = gather_error
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, m, T)
        if extras_child is not None:
            line["extra"] = collect_extras(extras_child)
            # the secondary numbers once more, compact and LAST in the line: a log that keeps only the tail of the line still shows them
            line["extra_summary"] = summarise_extras(line["extra"])
        # five keys FIRST in the line (a log that keeps only the head of the line still shows them), the full blocks follow
        sx = line.get("extra_summary") or {}
        first = {"summary": {"headline_frac_of_fp32_peak": round(achieved / PEAK_F32_TFLOPS, 4),
                             "strict_f32_variant_frac": None if f32_ms is None else round(line["strict_f32_variant"]["frac"], 4),
                             "lqr_mfma_option": line["lqr_mfma_option"],
                             "ilqr_api_warm_ms_frac": sx.get("ilqr_api_warm"),
                             "secondary_ms_frac": {k: sx.get(k) for k in ("cfg5_hvac", "cfg5_reservoir", "hvac6", "res4", "literal_dims") if k in sx}}}
        first.update(line)
        print(json.dumps(first), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
