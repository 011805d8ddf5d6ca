import os, sys
sys.path.insert(0, '/root/repo/tf-mpc_amd'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np, torch, problems
from tfmpc.solvers.lqr import LQR
np.set_printoptions(precision=4, suppress=True, linewidth=200)
F, f, C, c = problems.make_lqr_instance(1000, 16, 8)
x0 = np.random.default_rng(0).normal(size=16)
for T in (1, 2, 5):
    lqr = LQR(F, f, C, c)
    os.environ['TFMPC_LQR_KERNEL'] = 'generic'
    g = lqr.solve_device(x0, T, want_policy=True, want_value=True); torch.cuda.synchronize()
    os.environ['TFMPC_LQR_KERNEL'] = 'mfma'
    m = lqr.solve_device(x0, T, want_policy=True, want_value=True); torch.cuda.synchronize()
    print("T", T, "status", int(g['status'][0]), int(m['status'][0]))
    for key in ('K', 'k', 'V', 'v', 'const', 'states', 'actions', 'costs'):
        a, b = g[key][0].cpu().numpy(), m[key][0].cpu().numpy()
        print(key, "maxabs", np.abs(a).max(), "err", np.abs(a - b).max())
    if T == 1:
        print("K generic\n", g['K'][0, 0].cpu().numpy()[:3]); print("K mfma\n", m['K'][0, 0].cpu().numpy()[:3])
        print("k", g['k'][0, 0, :, 0].cpu().numpy(), "\n ", m['k'][0, 0, :, 0].cpu().numpy())
        print("V g\n", g['V'][0,0].cpu().numpy()[:3,:8]); print("V m\n", m['V'][0,0].cpu().numpy()[:3,:8])
