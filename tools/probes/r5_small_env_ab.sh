#!/bin/bash
# Round 5, the reference's own env configs (hvac6 / res4) at large batch: the tree's kernels against translation units 2 (HVAC, n <= 8) and 7 (Reservoir, n <= 4)
# of ilqr_adjoint_mfma.hip AS COMMITTED AT <rev> (default HEAD), linked against the product's other objects; both timed with tools/small_env_rates.py in one call.
#   tools/probes/r5_small_env_ab.sh build [rev]   (here)   then   tools/probes/r5_small_env_ab.sh   (GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/tools/probes/ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
if [ "$1" = build ]; then
  REV=${2:-HEAD}
  git -C $ROOT show $REV:tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip > $ROOT/tf-mpc_amd/csrc/.ab_rev.hip
  for P in 2 7; do
    /opt/rocm/bin/hipcc $FLAGS -DTFMPC_AM_PART=$P -x hip -c $ROOT/tf-mpc_amd/csrc/.ab_rev.hip -o $ROOT/tools/probes/ab/am${P}_rev.o 2>/dev/null &
  done
  wait
  rm -f $ROOT/tf-mpc_amd/csrc/.ab_rev.hip
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_small_rev.so $ROOT/tools/probes/ab/am2_rev.o $ROOT/tools/probes/ab/am7_rev.o \
      $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "ilqr_adjoint_mfma\.p[27]\.o")
  exit 0
fi
cd $ROOT
for rep in 1 2 3; do
for L in product lib_small_rev.so; do
  if [ $L = product ]; then unset TFMPC_LIB; else export TFMPC_LIB=$ROOT/tools/probes/ab/$L; fi
  echo "== $L"
  python tools/small_env_rates.py 2>&1 | grep "B="
done; done
