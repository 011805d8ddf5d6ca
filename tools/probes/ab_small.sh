#!/bin/bash
# A/B library builds on the small-env iLQR rates: tools/probes/ab_small.sh libA.so libB.so ...
for rep in 1 2; do
for L in "$@"; do
  cp tools/probes/ab/$L tf-mpc_amd/tfmpc/_lib/libtfmpc_hip.so
  echo "== $L"; python tools/small_env_rates.py 2>&1 | grep "B="
done; done
