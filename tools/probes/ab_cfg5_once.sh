#!/bin/bash
# A/B builds on the cfg5 single-solve timing: tools/probes/ab_cfg5_once.sh libA.so libB.so   (each run twice, interleaved)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for rep in 1 2; do
for L in "$@"; do
  export TFMPC_LIB=$ROOT/tools/probes/ab/$L      # tfmpc/_hip.py loads this build; the product library is never touched
  echo "== $L"; python $ROOT/tools/cfg5_once.py 2>&1 | grep containers
done; done
