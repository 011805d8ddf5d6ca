"""One-off: the direct oracle test of the 16-per-wave costate kernel (tests/test_ilqr_costate_mfma_oracle_gpu.py) on random
state dimensions -- every packing (four, two, one instance per matrix-core column; one and two tiles).
python tools/probes/fuzz_costate_oracle.py [cases]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np
from tfmpc import _hip
import test_ilqr_costate_mfma_oracle_gpu as t
t.MIN_DECIDED = 0.15          # HVAC's second iteration moves fewer actions on some shapes than the tests' fixed configs
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 14
rng = np.random.default_rng(4)
bad = 0
_hip.set_option("TFMPC_ILQR_KERNEL", "costate_mfma")
for case in range(cases):
    kind = ("reservoir", "hvac")[case % 2]
    n, T, B = int(rng.integers(2, 33)), int(rng.integers(5, 41)), int(rng.choice([5, 16, 23, 70]))
    try:
        t.test_costate_mfma_against_the_oracle(kind, n, T, B)
        print(f"case {case:2d} {kind} n={n} T={T} B={B}: ok", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"case {case:2d} {kind} n={n} T={T} B={B}: FAIL {str(e)[:300]}", flush=True)
    except Exception:
        bad += 1
        traceback.print_exc()
print("failures:", bad); sys.exit(1 if bad else 0)
