"""fuzz_costate.py case 192 (HVAC n = 10, T = 28, B = 1000, one iteration) looked at: how many instances differ between the wave kernel and the
16-per-wave kernel, by how much, and do the group forms (TFMPC_COSTATE_WAVES) agree with each other bit for bit?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch, problems
from tfmpc import _hip
from tfmpc.envs.hvac import HVAC
from tfmpc.solvers.ilqr import iLQR
rng = np.random.default_rng(2026)
for case in range(193):
    kind = "reservoir" if case % 3 else "hvac"
    n, T, B = int(rng.integers(1, 33)), int(rng.integers(1, 60)), int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 200, 1000, 5000]))
    its = int(rng.integers(1, 9))
    if kind == "hvac":
        x0 = rng.uniform(5.0, 30.0, size=(B, n, 1)).astype(np.float32); its = 1
    else:
        x0 = rng.uniform(20.0, 90.0, size=(B, n, 1)).astype(np.float32)
    atol = float(rng.choice([5e-3, 0.05]))
print("case", case, kind, n, T, B, its, atol)
env = HVAC.load(dict(problems.hvac_config(n, seed=case)))
s = iLQR(env, max_iterations=its, atol=atol)
u0 = s.random_actions(T, B, seed=case)
outs = {}
for kern, waves in (("wave", None), ("costate_mfma", None), ("costate_mfma", "1"), ("costate_mfma", "2"), ("costate_mfma", "4"), ("costate_mfma", "8")):
    with _hip.option("TFMPC_ILQR_KERNEL", kern), _hip.option("TFMPC_COSTATE_WAVES", waves):
        o = s.solve_device(x0, T, u_init=u0, trace_rows=4); torch.cuda.synchronize()
        outs[(kern, waves)] = {k: v.clone() for k, v in o.items() if torch.is_tensor(v) and k != "workspace"}
w = outs[("wave", None)]
for key, f in outs.items():
    d = (w["states"] - f["states"]).abs().amax(dim=(1, 2))
    scale = float(w["states"].abs().max())
    print(key, "instances beyond 1e-5 of the scale:", int((d > 1e-5 * scale).sum()), "max", float(d.max()), "scale", scale,
          "same bits as the default 16-per-wave form:", all(torch.equal(f[k], outs[("costate_mfma", None)][k]) for k in ("states", "actions", "costs", "iterations", "status")))
bad = int(torch.argmax((w["states"] - outs[("costate_mfma", None)]["states"]).abs().amax(dim=(1, 2))))
print("instance", bad, "alpha_index wave / mfma:", w["trace"][bad, :2, 5].tolist(), outs[("costate_mfma", None)]["trace"][bad, :2, 5].tolist(),
      "J:", w["trace"][bad, :2, 7].tolist(), outs[("costate_mfma", None)]["trace"][bad, :2, 7].tolist())
