#!/bin/bash
for L in "$@"; do
  export TFMPC_LIB=$PWD/tools/probes/ab/$L      # tfmpc/_hip.py loads this build; the product library is never touched
  echo "== $L"; python tools/phase_split.py 2>&1 | grep -v amdgpu.ids | head -3
done
