"""One named group of bench.py's workloads, launched the way bench.py launches them, for rocprofv3 (round 6):

    python tools/profile_workload.py <group>         groups: headline ilqr_api ilqr_api_full cfg5_hvac cfg5_reservoir hvac6 res4 lqr32 literal_dims cfg4 cfg4_one_launch box_stable box cfg2

Per workload of the group: PROFILE_WARM (default 40) untimed launches, then PROFILE_TIMED (default 10) launches inside ONE pair of events --
bench.py's protocol -- and one JSON line {"workload", "ms_per_launch", "warm", "timed", "iterations"}.  tools/pmc_workload.sh reads those lines
beside the kernel trace of the same process, so the profile's average duration (last PROFILE_TIMED dispatches of the kernel) and the
tool's own number stand side by side in the summary, and bench.py can check its line against both."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems, workloads
from tfmpc import _hip
from tfmpc.solvers.ilqr import iLQR
from tfmpc.solvers.lqr import LQR

WARM, TIMED = int(os.environ.get("PROFILE_WARM", 40)), int(os.environ.get("PROFILE_TIMED", 10))


def run(name, launch, warm=None, timed=None, note=None, units=None):
    warm, timed = WARM if warm is None else warm, TIMED if timed is None else timed
    out = launch(None)
    for _ in range(max(warm - 1, 0)):
        out = launch(out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(timed):
        out = launch(out)
    e1.record()
    torch.cuda.synchronize()
    line = {"workload": name, "ms_per_launch": e0.elapsed_time(e1) / timed, "warm": warm, "timed": timed}
    if isinstance(out, dict) and "iterations" in out:
        line["iterations"] = float((out["iterations"].double() + 1).sum())
        line["flagged"] = int((out["status"] != 0).sum())
    line["units_per_launch"] = units if units is not None else line.get("iterations")       # what bench.py scales the counters by
    if note:
        line["note"] = note
    print("PROFILE_WORKLOAD " + json.dumps(line), flush=True)


def ilqr_launcher(solver, x0, u0, T):
    return lambda out: solver.solve_device(x0, T, u_init=u0, workspace=None if out is None else out["workspace"])


def lqr_launcher(w):
    return lambda out: w["lqr"].solve_device(w["x0"], w["T"], workspace=None if out is None else out["workspace"])


def group_headline():
    run("headline_lqr_n16_m8_T50_B65536", lqr_launcher(workloads.headline(65536)), units=65536)


def group_ilqr_api():
    w = workloads.ilqr_api_warm(65536, 16, 8, 50)
    run("ilqr_api_warm", ilqr_launcher(workloads.solver_of(w), w["x0"], w["u0"], w["T"]))


def group_ilqr_api_full():
    w = workloads.ilqr_api_warm(65536, 16, 8, 50)
    with _hip.option("TFMPC_ILQR_LQ_REUSE", "0"):
        run("ilqr_api_warm_full_pass_every_iteration", ilqr_launcher(workloads.solver_of(w), w["x0"], w["u0"], w["T"]))


def _cfg5(kind):
    w = workloads.cfg5(kind, 32768)
    run(f"cfg5_{kind}", ilqr_launcher(iLQR(w["env"], max_iterations=12), w["x0"], w["u0"], w["T"]))


def group_cfg5_hvac(): _cfg5("hvac")
def group_cfg5_reservoir(): _cfg5("reservoir")


def _small(name):
    w = workloads.small_env(name)
    run(name, ilqr_launcher(w["solver"], w["x0"], w["u0"], w["T"]))


def group_hvac6(): _small("hvac6")
def group_res4(): _small("res4")


def group_lqr32():
    run("lqr_n32_m16_T50_B8192", lqr_launcher(workloads.lqr32()), units=8192)


def group_literal_dims():
    w = workloads.literal_dims(32768)
    run("literal_dims_ilqr_lq_n32_m16_T100_B32768", ilqr_launcher(workloads.solver_of(w), w["x0"], w["u0"], w["T"]), warm=min(WARM, 12), timed=min(TIMED, 6),
        note="37 ms per launch: 12 + 6 launches")


def group_cfg4():
    w = workloads.cfg4()
    run("cfg4_navigation_B16384", ilqr_launcher(w["solver"], w["x0"], w["u0"], w["T"]))


def group_cfg4_one_launch():
    w = workloads.cfg4(batches=8)
    run("cfg4_navigation_one_launch_of_8x16384", ilqr_launcher(w["solver"], w["x0"], w["u0"], w["T"]), warm=min(WARM, 12), timed=min(TIMED, 6),
        note="the persistent group kernel: grid = resident groups; 12 + 6 launches")


def group_box_stable():
    w = workloads.control_limited_stable(65536)
    run("control_limited_stable", ilqr_launcher(workloads.solver_of(w), w["x0"], w["u0"], w["T"]), warm=min(WARM, 6), timed=min(TIMED, 6),
        note="60 ms per launch: 6 + 6 launches")


def group_box():
    w = workloads.control_limited(65536)
    run("control_limited", ilqr_launcher(workloads.solver_of(w), w["x0"], w["u0"], w["T"]), warm=1, timed=2, note="0.5 s per launch: 1 + 2 launches")


def group_cfg2():
    run("cfg2_navlin_beta5", lqr_launcher(workloads.cfg2()), units=4096)


if __name__ == "__main__":
    globals()["group_" + sys.argv[1]]()
