"""Navigation (configs[3]: B = 16 384, T = 50) from PLAIN TORCH FUNCTIONS through TorchEnv.to_device_env() beside the hand-written DeviceEnv
source and the host-driven TorchEnv solve (on a sample): python tools/fxenv_rate.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np, torch
sys.argv = sys.argv[:1]
import bench
r = bench.deviceenv_rate()
print(json.dumps({"hand_written_Mit_s": r["iterations_per_s"] / 1e6, "hand_written_ms": r["ms_per_batch"], "from_python": r["from_python_functions"],
                  "builtin_lane_group_ms": r["same_env_builtin_lane_group_kernel_ms"]}, indent=1))
