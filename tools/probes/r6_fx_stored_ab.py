import os, sys, json
ROOT = '/root/repo'
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
sys.argv = sys.argv[:1]
import bench
for stored in ("4", "16", "4", "16"):
    os.environ["TFMPC_GROUP_STORED"] = stored
    r = bench.deviceenv_rate()
    print(stored, round(r["ms_per_batch"], 3), round(r["from_python_functions"]["ms_per_batch"], 3), round(r["same_env_builtin_lane_group_kernel_ms"], 3), flush=True)
