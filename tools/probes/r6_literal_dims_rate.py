"""literal dims (iLQR on the LQ env, n = 32, m = 16, T = 100, B = 32 768): ms per launch with gain reuse on / off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import torch, workloads
from tfmpc import _hip
w = workloads.literal_dims(int(os.environ.get("B", 32768)))
for mode in (None, "0", None):
    with _hip.option("TFMPC_ILQR_LQ_REUSE", mode):
        s = workloads.solver_of(w)
        o = s.solve_device(w["x0"], w["T"], u_init=w["u0"])
        for _ in range(3): o = s.solve_device(w["x0"], w["T"], u_init=w["u0"], workspace=o["workspace"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): o = s.solve_device(w["x0"], w["T"], u_init=w["u0"], workspace=o["workspace"])
        e1.record(); torch.cuda.synchronize()
    print("reuse", "on" if mode is None else "off", round(e0.elapsed_time(e1) / 5, 3), "ms; iterations", float((o["iterations"].double() + 1).sum()), "flagged", int((o["status"] != 0).sum()), s.last_kernel)
