#!/bin/bash
# A/B two builds on cfg5: tools/probes/ab_cfg5.sh libA.so libB.so
for L in "$@"; do
  cp tools/probes/ab/$L tf-mpc_amd/tfmpc/_lib/libtfmpc_hip.so
  echo "== $L"; python tools/secondary_rates.py 2>&1 | grep cfg5
done
