#!/bin/bash
# Probe build of the control-limited kernel (per-instance lifetime counters, tools/probes/box_lifetime.py): ilqr_lq_box_mfma.hip with
# -DTFMPC_BOX_PROBE, everything else from tf-mpc_amd/csrc/build -> tools/probes/ab/lib_boxprobe.so (loaded through TFMPC_LIB).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/tools/probes/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DTFMPC_BOX_PROBE $EXTRA \
    -c $ROOT/tf-mpc_amd/csrc/ilqr_lq_box_mfma.hip -o $ROOT/tools/probes/ab/boxprobe.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_boxprobe.so $ROOT/tools/probes/ab/boxprobe.o \
    $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "/ilqr_lq_box_mfma\.o")
