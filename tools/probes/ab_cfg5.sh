#!/bin/bash
# A/B two builds on cfg5: tools/probes/ab_cfg5.sh libA.so libB.so
for L in "$@"; do
  export TFMPC_LIB=$PWD/tools/probes/ab/$L      # tfmpc/_hip.py loads this build; the product library is never touched
  echo "== $L"; python tools/secondary_rates.py 2>&1 | grep cfg5
done
