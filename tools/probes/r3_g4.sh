#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
P="python tools/probes/group_waves_sweep.py"
$P hvac6 6 16384 2,4,8
$P res4 4 32768 2,4,8
$P reservoir 16 8192 2,4,8
$P hvac 16 8192 2,4,8
$P reservoir 32 8192 1,2,4,8
$P hvac 32 8192 1,2,4,8
$P hvac 12 6000,8192 2,4,8
