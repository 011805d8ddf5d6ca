#!/bin/bash
# Probe build of the library: ONE source of tf-mpc_amd/csrc recompiled with -DTFMPC_PHASE_PROBE (default lqr_block.hip),
# everything else from tf-mpc_amd/csrc/build -> tools/probes/ab/lib_probe.so (loaded through TFMPC_LIB, never copied over
# the product library).   tools/probes/build_probe.sh [ilqr_lane.hip]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
SRC=${1:-lqr_block.hip}
mkdir -p $ROOT/tools/probes/ab
make -j8 -C $ROOT/tf-mpc_amd/csrc > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DTFMPC_PHASE_PROBE -DTFMPC_CFG5_TRACE $PROBE_EXTRA \
    -c $ROOT/tf-mpc_amd/csrc/$SRC -o $ROOT/tools/probes/ab/probe.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_probe.so $ROOT/tools/probes/ab/probe.o \
    $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "/${SRC%.hip}\(\.p[0-9]\)\?\.o")
