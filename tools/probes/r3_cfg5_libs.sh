#!/bin/bash
# round 3: cfg5 single-solve timing of several builds: tools/probes/r3_cfg5_libs.sh libA.so libB.so ... ("product" = the tree's library)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for rep in 1 2; do
  for L in "$@"; do
    if [ "$L" = "product" ]; then unset TFMPC_LIB; else export TFMPC_LIB=$ROOT/tools/probes/ab/$L; fi
    echo "== $L"; python tools/cfg5_once.py 2>&1 | grep -E "containers"
  done
done
