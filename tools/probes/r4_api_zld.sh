#!/bin/bash
# Trajectory row stride of ilqr_lq_mfma_kernel in LDS: 26 floats (14.4 KB per wave: 11 waves per CU) against 24 (13.6 KB: 12 waves per CU = three on
# every SIMD; rows i and i + 8 then share banks).  Builds tools/probes/ab/lib_lq_zld24.so and times both with tools/probes/r4_api_occupancy.py.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/tools/probes/ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
for Z in 24 28; do
/opt/rocm/bin/hipcc $FLAGS -DTFMPC_LQ_ZLD=$Z -c $ROOT/tf-mpc_amd/csrc/ilqr_lq_mfma.hip -o $ROOT/tools/probes/ab/lq_zld$Z.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_lq_zld$Z.so $ROOT/tools/probes/ab/lq_zld$Z.o \
    $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "/ilqr_lq_mfma\.o")
done
for rep in 1 2; do
  for L in product lib_lq_zld24.so lib_lq_zld28.so; do
    if [ $L = product ]; then unset TFMPC_LIB; else export TFMPC_LIB=$ROOT/tools/probes/ab/$L; fi
    echo "$L: $(python $ROOT/tools/probes/r4_api_occupancy.py 2>/dev/null)"
  done
done
