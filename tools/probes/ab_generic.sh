#!/bin/bash
for L in "$@"; do
  cp tools/probes/ab/$L tf-mpc_amd/tfmpc/_lib/libtfmpc_hip.so
  echo "== $L"; python tools/generic_rates.py 2>&1 | grep -v amdgpu.ids
done
