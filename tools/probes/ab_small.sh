#!/bin/bash
# A/B library builds on the small-env iLQR rates: tools/probes/ab_small.sh libA.so libB.so ...
for rep in 1 2; do
for L in "$@"; do
  export TFMPC_LIB=$PWD/tools/probes/ab/$L      # tfmpc/_hip.py loads this build; the product library is never touched
  echo "== $L"; python tools/small_env_rates.py 2>&1 | grep "B="
done; done
