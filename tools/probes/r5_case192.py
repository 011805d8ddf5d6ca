"""fuzz_costate.py's case 192 (HVAC n = 10, T = 28, B = 1000, one iteration), compactly: how many instances differ between the wave kernel and the costate kernel, and by how much."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.solvers.ilqr import iLQR
target = 192
rng = np.random.default_rng(2026)
for case in range(target + 1):
    kind = "reservoir" if case % 3 else "hvac"
    n, T, B = int(rng.integers(1, 33)), int(rng.integers(1, 60)), int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 200, 1000, 5000]))
    its = int(rng.integers(1, 9))
    x0 = rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32) if kind == "hvac" else rng.uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)
    atol = float(rng.choice([5e-3, 0.05]))
env = HVAC.load(dict(problems.hvac_config(n, seed=case)))
s = iLQR(env, max_iterations=1, atol=atol); u0 = s.random_actions(T, B, seed=case)
out = {}
for kern in ("wave", "costate_mfma"):
    with _hip.option("TFMPC_ILQR_KERNEL", kern):
        o = s.solve_device(x0, T, u_init=u0, trace_rows=4); torch.cuda.synchronize()
        out[kern] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}
w, f = out["wave"], out["costate_mfma"]
d = (w["states"] - f["states"]).abs().flatten(1).amax(dim=1)
bad = torch.nonzero(d > 1e-5 * float(w["states"].abs().max())).flatten().tolist()
print(f"n={n} T={T} B={B}: {len(bad)} instances differ beyond 1e-5 of the scale: {bad[:8]}")
for b in bad[:3]:
    print("  instance", b, "wave trace", np.round(w["trace"][b, 0].cpu().numpy(), 6).tolist(), "\n            costate   ", np.round(f["trace"][b, 0].cpu().numpy(), 6).tolist())
