"""Where a cfg5 group's time goes (probe build: tools/probes/build_probe.sh ilqr_adjoint_mfma.hip, loaded through
TFMPC_LIB): s_memtime cycles per phase of ilqr_adjoint_mfma_kernel -- start rollout, costate sweeps, line-search passes,
stored rollouts -- per group, averaged; and the same at ONE group per SIMD (B = 16 384) and a single group (B = 16)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TFMPC_LIB", os.path.join(ROOT, "tools/probes/ab/lib_probe.so"))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.envs.reservoir import Reservoir
from tfmpc.solvers.ilqr import iLQR
n, T = 32, 100
lib = _hip.load()
lib.tfmpc_debug_cfg5_phases.argtypes = [ctypes.c_void_p]
names = ["start rollout", "sweeps", "search passes", "stored rollouts", "sweep coefficients", "wait: pass exchange", "wait: segments"]
SMALL = os.environ.get("PHASES_SMALL")                      # the reference's own configs (hvac6, res4) instead of n = 32
for B in [int(v) for v in sys.argv[1:]] or (32768, 16384, 16):
    rng = np.random.default_rng(4)
    for kind in (("hvac6", "res4") if SMALL else ("hvac", "reservoir")):
        if kind == "hvac6":
            env = HVAC.load(dict(problems.HVAC6_CONFIG)); x0 = np.tile(np.array(problems.HVAC6_X0, dtype=np.float32)[None], (B, 1, 1))
        elif kind == "res4":
            env = Reservoir.load(dict(problems.RES4_CONFIG)); x0 = np.tile(np.array(problems.RES4_X0, dtype=np.float32)[None], (B, 1, 1))
        elif kind == "hvac":
            env = HVAC.load(dict(problems.hvac_config(n, seed=5))); x0 = np.full((B, n, 1), 10.0, dtype=np.float32)
        else:
            env = Reservoir.load(dict(problems.reservoir_config(n, seed=5))); x0 = rng.uniform(50, 75, size=(B, n, 1)).astype(np.float32)
        s = iLQR(env, max_iterations=12); u0 = s.random_actions(T, B, seed=5)
        per = 16 * (4 if env.state_size <= 4 else (2 if env.state_size <= 8 else 1))
        groups = (B + per - 1) // per
        buf = torch.zeros((groups, 8, 16), dtype=torch.int64, device="cuda")       # [group][wave of the group][slot]
        assert lib.tfmpc_debug_cfg5_phases(buf.data_ptr()) == 0
        out = s.solve_device(x0, T, u_init=u0); torch.cuda.synchronize()
        out = s.solve_device(x0, T, u_init=u0, workspace=out["workspace"]); torch.cuda.synchronize()
        pw = buf.cpu().numpy().astype(np.float64)
        p = pw[:, 0]                                       # wave 0 runs every phase
        tot = p[:, 5].mean()
        print(f"{kind} B={B}: kernel {tot / 1e3:.0f} k ticks per group (s_memtime ticks)")
        for i, nm in enumerate(names):
            c = p[:, 8 + i].mean()
            i = 7 if i == 5 else i          # (slot 5 holds the kernel's total)
            print(f"   {nm:16s} {100 * p[:, i].mean() / tot:5.1f} %   {c:6.1f} calls, {p[:, i].mean() / max(c, 1):9.0f} ticks per call")
        nw = int((pw[:, :, 5].mean(axis=0) > 0).sum())
        if nw > 1:                                         # multi-wave groups: the line-search rollouts wave by wave
            print("   search rollouts by wave: " + ", ".join(f"{pw[:, w, 2].mean() / max(pw[:, w, 10].mean(), 1) / 1e3:.0f} k x {pw[:, w, 10].mean():.1f}" for w in range(nw)))
            print("   stored segments by wave: " + ", ".join(f"{pw[:, w, 3].mean() / max(pw[:, w, 11].mean(), 1) / 1e3:.1f} k" for w in range(nw)))
