"""Per-phase cycle counts of one iLQR iteration in ilqr_group_solve_kernel (cfg4: Navigation, n = m = 2, T = 50).  Needs
the probe build: tools/probes/build_probe.sh ilqr_lane.hip.  atol = 0, so every instance runs max_iterations.
Run on the GPU box: python tools/probes/lane_probe.py [B]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TFMPC_LIB"] = f'{root}/tools/probes/ab/lib_probe.so'
sys.path.insert(0, f'{root}/tf-mpc_amd'); sys.path.insert(0, f'{root}/tests')
import numpy as np, torch, problems
from tfmpc.envs.navigation import Navigation
from tfmpc.solvers.ilqr import iLQR
B, T, its = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 50, 20
s = iLQR(Navigation.load(problems.NAV_CONFIG), atol=0.0, max_iterations=its)
x0 = np.random.default_rng(4).uniform(0, 10, size=(B, 2, 1)).astype(np.float32)
out = s.solve_device(x0, T, u_init=s.random_actions(T, B, seed=4)); torch.cuda.synchronize()
acc = out["costs"][0, :8].cpu().numpy().astype(np.float64)
n_it = float(out["iterations"][0]) + 1
names = ["linearize", "Q matrices", "box-QP", "K solve", "V update + stores", "speculative rollout", "replay"]
print(f"B = {B}: instance 0 ran {n_it:.0f} iterations; s_memtime ticks (100 MHz) per iteration")
for nm, v in zip(names, acc): print(f"  {nm:22s} {v / n_it:10.0f}")
print(f"  {'sum':22s} {acc.sum() / n_it:10.0f}")
