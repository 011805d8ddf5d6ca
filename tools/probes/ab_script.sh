#!/bin/bash
# A/B builds on any tool script: tools/probes/ab_script.sh "<script.py args>" libA.so libB.so   (each run twice, interleaved)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
CMD=$1; shift
for rep in 1 2; do
for L in "$@"; do
  export TFMPC_LIB=$ROOT/tools/probes/ab/$L      # tfmpc/_hip.py loads this build; the product library is never touched
  echo "== $L"; (cd $ROOT && python $CMD 2>&1 | grep -v amdgpu.ids)
done; done
