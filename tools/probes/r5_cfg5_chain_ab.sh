#!/bin/bash
# Register budget / step sizes per pass / ring depth of the Reservoir chain instantiation (ilqr_adjoint_mfma.hip part 8): variant builds of that ONE
# translation unit linked against the product's other objects, each timed against the general kernel in the same process.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/tools/probes/ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DTFMPC_AM_PART=8"
build() {   # name, extra flags
/opt/rocm/bin/hipcc $FLAGS $2 -c $ROOT/tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip -o $ROOT/tools/probes/ab/am8_$1.o 2>/dev/null
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_am8_$1.so $ROOT/tools/probes/ab/am8_$1.o \
    $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "ilqr_adjoint_mfma\.p8\.o")
}
if [ "$1" = build ]; then
build eu3a2 "-DTFMPC_CHAIN_ALPHAS=2"
build eu2a2 "-DTFMPC_TWO_TILE_EU(K)=2 -DTFMPC_CHAIN_ALPHAS=2"
build eu2a2r3 "-DTFMPC_TWO_TILE_EU(K)=2 -DTFMPC_CHAIN_ALPHAS=2 -DTFMPC_CHAIN_RING=3"
build eu3a1r3 "-DTFMPC_CHAIN_RING=3"
exit 0
fi
cd $ROOT
for L in product lib_am8_eu3a2.so lib_am8_eu2a2.so lib_am8_eu2a2r3.so lib_am8_eu3a1r3.so; do
  if [ $L = product ]; then unset TFMPC_LIB; else export TFMPC_LIB=$ROOT/tools/probes/ab/$L; fi
  echo "== $L (product = three waves per SIMD, one step size per pass, ring depth 2)"
  python tools/probes/r5_cfg5_chain_ab.py 32768 8192
done
