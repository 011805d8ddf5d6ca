#!/bin/bash
# Probe build of the library: lqr_block.hip with -DTFMPC_PHASE_PROBE, everything else from tf-mpc_amd/csrc/build.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/tools/probes/ab
make -C $ROOT/tf-mpc_amd/csrc > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DTFMPC_PHASE_PROBE \
    -c $ROOT/tf-mpc_amd/csrc/lqr_block.hip -o $ROOT/tools/probes/ab/lqr_block_probe.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_probe.so $ROOT/tools/probes/ab/lqr_block_probe.o \
    $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v lqr_block.o)
