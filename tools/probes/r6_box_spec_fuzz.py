"""Fuzz of the speculative sweeps of the control-limited kernel's helper teams (ilqr_lq_box_mfma.hip, round 6): random batch sizes above 4 096,
shapes, horizons, bounds, team counts, claim thresholds and speculation thresholds; every output of the launch with speculating teams must equal the
launch without helpers (TFMPC_BOX_HELPERS=off) bit for bit.  No decision trace (a traced solve does not speculate); the number of rejected passes
comes from a separate traced launch.     python tools/probes/r6_box_spec_fuzz.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.lq import LQEnv
from tfmpc.solvers.ilqr import iLQR
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 21)
bad = 0
t_start = time.time()
for case in range(cases):
    n = int(rng.integers(3, 17)); m = int(rng.integers(max(1, 7 - n), 9)); T = int(rng.integers(4, 61))
    B = int(rng.integers(4097, 7000)); bound = float(rng.choice([0.05, 0.1, 0.5, 2.0]))
    scale = float(rng.choice([0.18, 0.2, 0.22]))
    teams = int(rng.choice([1, 2, 5, 16])); after = int(rng.choice([0, 1, 2, 4])); spec = int(rng.choice([0, 0, 1, 2]))
    F, f, C, c, x0 = problems.make_lqr_batch_fast(B, n, m, seed=int(rng.integers(1 << 30)))
    F = F * scale * np.sqrt(16.0 / n)
    x0 = x0 * float(rng.choice([1.0, 3.0, 10.0]))                 # (larger starts: more clamping, more rejected passes)
    solver = iLQR(LQEnv(F, f, C, c, low=-bound, high=bound), max_iterations=int(rng.integers(3, 25)))
    x0d = torch.as_tensor(x0[..., None].astype(np.float32), device="cuda")
    u0 = torch.zeros(B, T, m, 1, device="cuda")
    outs = {}
    for mode in ("off", str(teams)):
        with _hip.option("TFMPC_BOX_HELPERS", mode), _hip.option("TFMPC_BOX_HELP_AFTER", str(after)), _hip.option("TFMPC_BOX_SPECULATE", str(spec)):
            o = solver.solve_device(x0d, T, u_init=u0)
            torch.cuda.synchronize()
            outs[mode] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}
    tr = solver.solve_device(x0d, T, u_init=u0, trace_rows=64)
    torch.cuda.synchronize()
    valid = torch.arange(64, device="cuda")[None, :] < tr["trace_len"][:, None]
    rejected = int(((tr["trace"][..., 8] == 0) & valid).sum())
    same = all(torch.equal(outs["off"][k], outs[str(teams)][k]) for k in ("states", "actions", "costs", "iterations", "status"))
    bad += 0 if same else 1
    print(f"case {case}: B {B} n {n} m {m} T {T} bound {bound} F x {scale} teams {teams} after {after} speculate {spec} rejected passes {rejected} "
          f"flagged {int((outs['off']['status'] != 0).sum())}: {'same bits' if same else 'MISMATCH'}", flush=True)
print(f"{cases} cases, {bad} mismatches, {time.time() - t_start:.0f} s")
sys.exit(1 if bad else 0)
