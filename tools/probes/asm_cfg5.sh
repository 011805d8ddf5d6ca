#!/bin/bash
# Device assembly of ONLY the cfg5 instantiations (n = 32: two tiles, 16-byte pieces, fp32 containers) of
# tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip, for register / instruction budgeting without a GPU:
#   tools/probes/asm_cfg5.sh [out.s]     then  python tools/probes/asm_loops.py out.s <kernel substring>
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${1:-/tmp/asm/cfg5.s}
mkdir -p $(dirname $OUT)
# (the file is built as eight translation units: part 0 = HVAC, two tiles, part 4 = Reservoir, two tiles -- every VW / container / group form)
for P in 0 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DTFMPC_AM_PART=$P -S --cuda-device-only \
      -o $OUT.p$P $ROOT/tf-mpc_amd/csrc/ilqr_adjoint_mfma.hip 2>&1 | grep -v "warning: argument unused" | grep -v "pass-failed\|__global__\|\^\|warnings generated" || true
done
cat $OUT.p0 $OUT.p4 > $OUT; rm -f $OUT.p0 $OUT.p4
python3 - $OUT <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S*ilqr_adjoint_mfma_kernel\S*)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', txt):
    t = re.search(r'ILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELi(\d)E', m.group(1))
    print("KIND", t.group(1), "NT", t.group(2), "NW", t.group(6), "| vgpr", m.group(3), "vgpr spills", m.group(4), "sgpr spills", m.group(2))
PY
