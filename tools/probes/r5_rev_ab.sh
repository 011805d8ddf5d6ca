#!/bin/bash
# The tree's costate kernels against translation units of ilqr_adjoint_mfma.hip AS COMMITTED AT <rev>, linked with the product's other objects,
# alternating in one call on one box:
#   [AB_FLAGS=...] tools/probes/r5_rev_ab.sh build <rev | WORK> <part> [<part> ...]     (here; parts 0..8, see the file; WORK = the working tree, for flag A/Bs)
#   tools/probes/r5_rev_ab.sh run <python tool> [grep pattern]     (GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $ROOT/tools/probes/ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
if [ "$1" = build ]; then
  REV=$2; shift 2
  if [ "$REV" = WORK ]; then cp $ROOT/tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip $ROOT/tf-mpc_amd/csrc/.ab_rev.hip      # the working tree, with $AB_FLAGS
  else git -C $ROOT show $REV:tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip > $ROOT/tf-mpc_amd/csrc/.ab_rev.hip; fi
  objs=""; excl=""
  for P in "$@"; do
    /opt/rocm/bin/hipcc $FLAGS $AB_FLAGS -DTFMPC_AM_PART=$P -x hip -c $ROOT/tf-mpc_amd/csrc/.ab_rev.hip -o $ROOT/tools/probes/ab/rev_p$P.o 2>/dev/null &
    objs="$objs $ROOT/tools/probes/ab/rev_p$P.o"; excl="$excl\|ilqr_adjoint_mfma\.p$P\.o"
  done
  wait
  rm -f $ROOT/tf-mpc_amd/csrc/.ab_rev.hip
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_rev.so $objs $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "XXXX$excl")
  exit 0
fi
if [ "$1" = buildfile ]; then      # tools/probes/r5_rev_ab.sh buildfile <rev> <source.hip>: ONE other source of csrc as committed at <rev>
  REV=$2; SRC=$3
  if [ "$REV" = WORK ]; then cp $ROOT/tf-mpc_amd/csrc/$SRC $ROOT/tf-mpc_amd/csrc/.ab_rev.hip; else git -C $ROOT show $REV:tf-mpc_amd/csrc/$SRC > $ROOT/tf-mpc_amd/csrc/.ab_rev.hip; fi
  /opt/rocm/bin/hipcc $FLAGS $AB_FLAGS -x hip -c $ROOT/tf-mpc_amd/csrc/.ab_rev.hip -o $ROOT/tools/probes/ab/rev_file.o 2>/dev/null
  rm -f $ROOT/tf-mpc_amd/csrc/.ab_rev.hip
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/probes/ab/lib_rev.so $ROOT/tools/probes/ab/rev_file.o $(ls $ROOT/tf-mpc_amd/csrc/build/*.o | grep -v "/${SRC%.hip}\.o")
  exit 0
fi
cd $ROOT
TOOL=$2; PAT=${3:-.}
for rep in 1 2 3; do
for L in product lib_rev.so; do
  if [ $L = product ]; then unset TFMPC_LIB; else export TFMPC_LIB=$ROOT/tools/probes/ab/$L; fi
  echo "== $L"
  python $TOOL 2>&1 | grep "$PAT"
done; done
