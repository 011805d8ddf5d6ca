#!/bin/bash
# Register / scratch use of every kernel in one HIP source of tf-mpc_amd/csrc (compiles it alone with the Makefile's flags).
#   tools/probes/kernel_regs.sh ilqr_adjoint_mfma.hip
set -e
src=$(cd "$(dirname "$0")/../.." && pwd)/tf-mpc_amd/csrc/$1
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -c "$src" -o "$tmp/x.o" -save-temps=obj 2>&1 | grep -v warning | head
grep -E "^\s+\.(vgpr_count|private_segment_fixed_size|vgpr_spill_count|sgpr_count):|^\s+\.name:\s+_Z" "$tmp"/*gfx950.s | sed 's/\s\+/ /g' | paste - - - - - | sed 's/_ZN5tfmpc12_GLOBAL__N_1//;s/EEEv8Tfmpc[A-Za-z0-9_]*//' | cut -c1-220
cp "$tmp"/*gfx950.s /tmp/last_kernel.s
rm -rf "$tmp"
