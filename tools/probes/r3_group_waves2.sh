#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
P="python tools/probes/group_waves_sweep.py"
$P hvac6 6 1024,4096,8192,16384,32768,65536
$P res4 4 1024,4096,8192,16384,32768,65536,131072
$P hvac 32 16,256,1024,4096 1 default
$P hvac 32 16,256,1024,4096 1,2,4,8 costate_mfma
$P hvac 32 8192,12288,16384
$P reservoir 32 16,256,8192,12288,16384
$P hvac 16 256,4096,8192,16384
$P reservoir 16 256,4096,8192,16384
