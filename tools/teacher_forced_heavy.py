"""Every pass of the heavy control-limited instances, teacher-forced on the device's box-QP free sets, in two halves:

    python tools/teacher_forced_heavy.py dump gpurun_out/r06_teacher_forced_device.pkl      (GPU box: seconds)
    python tools/teacher_forced_heavy.py replay gpurun_out/r06_teacher_forced_device.pkl profiles/r06_teacher_forced_control_limited.txt [workers]
                                                                                              (any machine: minutes of restatement on CPU cores)

The device half records, for 16 instances in order, 4 with Cholesky retries, 4 of the 100-iteration family and 4 at the attempt cap (the
sampled in-suite test covers more of the light ones): the
decision trace, the clamp mask and iteration count of every box-QP (tfmpc_ilqr_solve_trace_qp_f32), the nominal trajectory at the start of
every iteration.  The replay half hands every sampled pass -- EVERY pass of four instances of each heavy group -- to the fp32 / fp64
restatements with K_t forced onto the device's free sets and asserts what tests/test_ilqr_teacher_forced_gpu.py asserts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import test_ilqr_teacher_forced_gpu as t

if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    if mode == "dump":
        t._control_limited(n_order=16, n_group=4, cap_light=24, cap_heavy=12, every_pass_of=4, dump=path)
        print("wrote", path, os.path.getsize(path), "bytes")
    else:
        stats = t._control_limited(0, 0, 0, 0, replay=path, log=sys.argv[3] if len(sys.argv) > 3 else None, workers=int(sys.argv[4]) if len(sys.argv) > 4 else None)
        print(dict(stats))
