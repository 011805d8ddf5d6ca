#!/bin/bash
# round 3: the reference's own env configs (hvac6 / res4, B = 16 384) and n = 32 at small batches, per group form
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for W in 1 2 4 8; do
  echo "== TFMPC_COSTATE_WAVES=$W"; TFMPC_COSTATE_WAVES=$W python tools/small_env_once.py 2>&1 | grep -E " ms"
done
for B in 1024 4096 8192; do
  for W in 1 2 4 8; do
    echo "== n = 32, B = $B, TFMPC_COSTATE_WAVES=$W"; TFMPC_COSTATE_WAVES=$W CFG5_B=$B python tools/cfg5_once.py 2>&1 | grep -E "containers"
  done
done
