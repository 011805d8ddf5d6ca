#!/bin/bash
# Register budget of the persistent lane-group kernel: sized for 4 waves per SIMD (128 VGPRs, 19 spilled; the default) against 3 (168 VGPRs)
# and 2: builds tools/probes/ab/lib_lane_eu{3,2}.so and times tools/cfg4_one_launch.py with each.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/tools/probes/ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
for E in 2 1; do
/opt/rocm/bin/hipcc $FLAGS -DTFMPC_GROUP_LANE_EU=$E -c $ROOT/tf-mpc_amd/csrc/ilqr_lane.hip -o $ROOT/tools/probes/ab/lane_eu$E.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_lane_eu$E.so $ROOT/tools/probes/ab/lane_eu$E.o \
    $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "/ilqr_lane\.o")
done
for rep in 1 2 3; do
  for L in product lib_lane_eu2.so lib_lane_eu1.so; do
    if [ $L = product ]; then unset TFMPC_LIB; else export TFMPC_LIB=$ROOT/tools/probes/ab/$L; fi
    echo "$L: $(python $ROOT/tools/cfg4_one_launch.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(v['ms_median'],2), round(v['iterations_per_s_at_median']/1e6,1)) for k,v in d.items()})")"
  done
done
