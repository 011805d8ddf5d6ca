#!/bin/bash
# round 3: cfg5 timing of the round-2 kernel build vs the current tree (one / two waves per group)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for rep in 1 2; do
  echo "== r02 build"; TFMPC_LIB=$ROOT/tools/probes/ab/lib_r02.so python tools/cfg5_once.py 2>&1 | grep -E "containers|mean"
  echo "== current, one wave per group"; TFMPC_COSTATE_WAVES=1 python tools/cfg5_once.py 2>&1 | grep -E "containers|mean"
  if [ "$1" = "pair" ]; then echo "== current, two waves per group"; TFMPC_COSTATE_WAVES=2 python tools/cfg5_once.py 2>&1 | grep -E "containers|mean"; fi
done
