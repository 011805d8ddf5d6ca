"""One named group of bench.py's workloads, launched the way bench.py launches them, for rocprofv3 (round 6):

    python tools/profile_workload.py <group>         groups: headline ilqr_api cfg5 small_env large_tile cfg4 box_stable box cfg2

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


def run(name, launch, warm=None, timed=None, note=None):
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
    if note:
        line["note"] = note
    print("PROFILE_WORKLOAD " + json.dumps(line), flush=True)


def ilqr_launcher(solver, x0, u0, T):
    return lambda out: solver.solve_device(x0, T, u_init=u0, workspace=None if out is None else out["workspace"])


def group_headline():
    B, n, m, T = 65536, 16, 8, 50
    F, f, C, c, x0 = problems.make_lqr_batch_spd(B, n, m, seed=0)           # bench.py's headline generator
    lqr = LQR(F, f, C, c)
    x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
    run("headline_lqr_n16_m8_T50_B65536", lambda out: lqr.solve_device(x0d, T, workspace=None if out is None else out["workspace"]))


def group_ilqr_api():
    w = workloads.ilqr_api_warm(65536, 16, 8, 50)
    run("ilqr_api_warm", ilqr_launcher(workloads.solver_of(w), w["x0"], w["u0"], w["T"]))
    with _hip.option("TFMPC_ILQR_LQ_REUSE", "0"):
        run("ilqr_api_warm_full_pass_every_iteration", ilqr_launcher(workloads.solver_of(w), w["x0"], w["u0"], w["T"]), warm=10, timed=5)


def group_cfg5():
    for kind in ("hvac", "reservoir"):
        w = workloads.cfg5(kind, 32768)
        run(f"cfg5_{kind}", ilqr_launcher(iLQR(w["env"], max_iterations=12), w["x0"], w["u0"], w["T"]))


def group_small_env():
    from tfmpc.envs.hvac import HVAC
    from tfmpc.envs.reservoir import Reservoir
    B, T = 16384, 100
    for name, env, x0r in (("hvac6", HVAC.load(dict(problems.HVAC6_CONFIG)), problems.HVAC6_X0), ("res4", Reservoir.load(dict(problems.RES4_CONFIG)), problems.RES4_X0)):
        x0 = torch.as_tensor(np.tile(np.array(x0r, dtype=np.float32)[None], (B, 1, 1)), device="cuda")
        s = iLQR(env, max_iterations=12)
        run(name, ilqr_launcher(s, x0, s.random_actions(T, B, seed=1), T))


def group_large_tile():
    n, m, B = 32, 16, 8192
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=1)
    lqr = LQR(F * 0.5, f, C, c)
    x0d = lqr._prep_x0(x0)
    run("lqr_n32_m16_T50_B8192", lambda out: lqr.solve_device(x0d, 50, workspace=None if out is None else out["workspace"]))
    w = workloads.literal_dims(32768)
    run("literal_dims_ilqr_lq_n32_m16_T100_B32768", ilqr_launcher(workloads.solver_of(w), w["x0"], w["u0"], w["T"]), warm=10, timed=5)


def group_cfg4():
    from tfmpc.envs.navigation import Navigation
    T = 50
    solver = iLQR(Navigation.load(problems.NAV_CONFIG))
    for B in (16384, 131072):
        x0 = torch.as_tensor(np.concatenate([np.random.default_rng(4 if B == 16384 else 100 + i).uniform(0, 10, size=(16384, 2, 1)) for i in range(B // 16384)]).astype(np.float32), device="cuda")
        u0 = torch.cat([solver.random_actions(T, 16384, seed=4 if B == 16384 else 100 + i) for i in range(B // 16384)])
        run(f"cfg4_navigation_B{B}", ilqr_launcher(solver, x0, u0, T), warm=min(WARM, 20), note="one launch; the kernel is persistent: grid = resident groups")


def group_box_stable():
    w = workloads.control_limited_stable(65536)
    run("control_limited_stable", ilqr_launcher(workloads.solver_of(w), w["x0"], w["u0"], w["T"]), warm=min(WARM, 6), timed=min(TIMED, 6),
        note="60 ms per launch: 6 + 6 launches")


def group_box():
    w = workloads.control_limited(65536)
    run("control_limited", ilqr_launcher(workloads.solver_of(w), w["x0"], w["u0"], w["T"]), warm=1, timed=2, note="0.5 s per launch: 1 + 2 launches")


def group_cfg2():
    B, T = 4096, 50
    for beta in (5.0, 0.0):
        F, f, C, c, x0 = problems.make_navlin_batch(B, beta)
        lqr = LQR(F, f, C, c)
        x0d = lqr._prep_x0(x0)
        run(f"cfg2_navlin_beta{beta:g}", lambda out: lqr.solve_device(x0d, T, workspace=None if out is None else out["workspace"]))


if __name__ == "__main__":
    globals()["group_" + sys.argv[1]]()
