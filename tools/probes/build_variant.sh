#!/bin/bash
# Variant build of the library: ONE source of tf-mpc_amd/csrc recompiled with extra -D flags, everything else from tf-mpc_amd/csrc/build
#   tools/probes/build_variant.sh NAME SOURCE.hip -DFLAG ...   -> tools/probes/ab/lib_NAME.so   (load with TFMPC_LIB=...; the product is untouched)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
NAME=$1; SRC=$2; shift 2
mkdir -p $ROOT/tools/probes/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form "$@" \
    -c $ROOT/tf-mpc_amd/csrc/$SRC -o $ROOT/tools/probes/ab/$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_$NAME.so $ROOT/tools/probes/ab/$NAME.o \
    $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "/${SRC%.hip}\(\.p[0-9]\)\?\.o")
rm -f $ROOT/tools/probes/ab/$NAME.o
