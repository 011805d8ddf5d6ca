"""One launch (after a warm-up) of every kernel family at a representative size, for a single rocprofv3 --kernel-trace
--stats table (profiles/): headline LQR, iLQR API, block LQR, generic LQR, cfg2, cfg4, cfg5, hvac6 / res4 packed."""
import os, runpy
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for script in ("tools/secondary_rates.py", "tools/small_env_rates.py", "tools/generic_phase_split.py"):
    runpy.run_path(os.path.join(root, script), run_name="__main__")
