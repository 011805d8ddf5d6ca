#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for rep in 1 2; do
for L in product lib_eu2.so; do
  if [ "$L" = "product" ]; then unset TFMPC_LIB; else export TFMPC_LIB=$ROOT/tools/probes/ab/$L; fi
  echo "== $L"
  python tools/probes/group_waves_sweep.py hvac6 6 16,16384 auto 2>&1 | grep -v amdgpu
  python tools/probes/group_waves_sweep.py res4 4 16,16384 auto 2>&1 | grep -v amdgpu
  python tools/probes/group_waves_sweep.py reservoir 16 256,8192 auto 2>&1 | grep -v amdgpu
  python tools/probes/group_waves_sweep.py hvac 16 256,8192 auto 2>&1 | grep -v amdgpu
done
done
