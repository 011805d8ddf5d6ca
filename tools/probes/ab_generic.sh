#!/bin/bash
# A/B library builds on the fallback-kernel rates: tools/probes/ab_generic.sh libA.so libB.so ...
for rep in 1 2; do
for L in "$@"; do
  cp tools/probes/ab/$L tf-mpc_amd/tfmpc/_lib/libtfmpc_hip.so
  echo "== $L"; python tools/generic_rates.py 2>&1 | tail -2
done; done
