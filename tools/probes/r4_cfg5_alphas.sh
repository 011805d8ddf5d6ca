#!/bin/bash
# Step sizes per line-search pass of the two-tile Reservoir kernel now that its coupling product is a row shift (no matrix work, fewer live
# registers): 2 (the default since round 2) against 3 and 1.  Builds translation unit 4 (Reservoir, two tiles) with -DTFMPC_SEARCH_ALPHAS=n.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/tools/probes/ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
for A in 3 1; do
/opt/rocm/bin/hipcc $FLAGS -DTFMPC_AM_PART=4 -DTFMPC_SEARCH_ALPHAS=$A -c $ROOT/tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip -o $ROOT/tools/probes/ab/am4_a$A.o &
done
wait
for A in 3 1; do
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_res_a$A.so $ROOT/tools/probes/ab/am4_a$A.o \
    $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "/ilqr_adjoint_mfma\.p4\.o")
done
cd $ROOT
export CFG5_ONCE_SINGLE=
for rep in 1 2; do
  for L in product lib_res_a3.so lib_res_a1.so; do
    if [ "$L" = "product" ]; then unset TFMPC_LIB; else export TFMPC_LIB=$ROOT/tools/probes/ab/$L; fi
    echo "== $L: $(python tools/cfg5_once.py 2>&1 | grep -E "reservoir fp32 containers")"
  done
done
