#!/bin/bash
# tools/probes/r6_ab_phases.sh lib_a.so lib_b.so ...: r6_reuse_phases.py per variant build ("product" = the product library)
for L in "$@"; do
  if [ "$L" = product ]; then unset TFMPC_LIB; else export TFMPC_LIB=$PWD/tools/probes/ab/$L; fi
  echo -n "$L: "; python tools/probes/r6_reuse_phases.py 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print({k:v[0] for k,v in d.items() if 'on' in k or k=='cap1_reuseoff'})"
done
