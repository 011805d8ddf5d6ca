#!/bin/bash
# Does a fourth wave per SIMD pay in ilqr_lq_mfma_kernel?  Builds the kernel with amdgpu_waves_per_eu(4,4) (128 VGPRs) into
# tools/probes/ab/lib_lq_eu4.so and times both builds at T = 20 (LDS 7.9 KB per wave: 20 waves per CU fit, so the register budget decides)
# and T = 50 (14.4 KB: 11 waves per CU whatever the registers).    bash tools/probes/r4_api_occupancy.sh   (GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/tools/probes/ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc $FLAGS -DTFMPC_LQ_EU=4 -c $ROOT/tf-mpc_amd/csrc/ilqr_lq_mfma.hip -o $ROOT/tools/probes/ab/lq_eu4.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_lq_eu4.so $ROOT/tools/probes/ab/lq_eu4.o \
    $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "/ilqr_lq_mfma\.o")
for rep in 1 2; do
  for L in product lib_lq_eu4.so; do
    if [ $L = product ]; then unset TFMPC_LIB; else export TFMPC_LIB=$ROOT/tools/probes/ab/$L; fi
    echo -n "$L: "; python $ROOT/tools/probes/r4_api_occupancy.py
  done
done
