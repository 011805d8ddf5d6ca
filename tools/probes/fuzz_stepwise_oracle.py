"""One-off: the per-iteration oracle parity test of the fused iLQR kernels (tests/test_ilqr_stepwise_oracle_gpu.py) on random
LQ-env shapes, bounded and unbounded, both cost spectra.  python tools/probes/fuzz_stepwise_oracle.py [cases]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ('tf-mpc_amd', 'tests', ''): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np
import test_ilqr_stepwise_oracle_gpu as t
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(8)
bad = 0
for case in range(cases):
    n, m, T = int(rng.integers(5, 33)), int(rng.integers(1, 17)), int(rng.integers(4, 25))
    bound = None if case % 2 else float(rng.choice([0.3, 0.8, 2.0]))
    spectrum = ("narrow", "reference")[int(rng.integers(2))]
    try:
        t.test_lq_env_every_iteration(n, m, T, bound, spectrum)
        print(f"case {case:2d} n={n} m={m} T={T} bound={bound} {spectrum}: ok", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"case {case:2d} n={n} m={m} T={T} bound={bound} {spectrum}: FAIL {str(e)[:300]}", flush=True)
    except Exception:
        bad += 1
        traceback.print_exc()
print("failures:", bad); sys.exit(1 if bad else 0)
