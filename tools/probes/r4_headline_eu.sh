#!/bin/bash
# Register budget / LDS slice of the headline kernel: the compiler's own choice (108 VGPRs -> 4 waves per SIMD) against budgets sized for 5 (<= 96)
# and 6 (<= 80; the rollout chunk cut to 25 steps so that 24 waves fit a CU's LDS).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/tools/probes/ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
build() {   # name, extra flags
/opt/rocm/bin/hipcc $FLAGS $2 -c $ROOT/tf-mpc_amd/csrc/lqr_mfma16x8.hip -o $ROOT/tools/probes/ab/lqr_$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_lqr_$1.so $ROOT/tools/probes/ab/lqr_$1.o \
    $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "/lqr_mfma16x8\.o")
}
build eu5 "-DTFMPC_LQR_EU=5"
build eu5tc25 "-DTFMPC_LQR_EU=5 -DTFMPC_LQR_TC=25"
build eu6tc25 "-DTFMPC_LQR_EU=6 -DTFMPC_LQR_TC=25"
cd $ROOT
for rep in 1 2 3; do
  for L in product lib_lqr_eu5.so lib_lqr_eu5tc25.so lib_lqr_eu6tc25.so; do
    if [ $L = product ]; then unset TFMPC_LIB; else export TFMPC_LIB=$ROOT/tools/probes/ab/$L; fi
    echo "$L: $(python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['status_flagged_instances'])")"
  done
done
