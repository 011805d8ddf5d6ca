#!/bin/bash
# A/B two builds of the library on the same box: tools/probes/ab.sh libA.so libB.so
for rep in 1 2; do
for L in "$@"; do
  export TFMPC_LIB=$PWD/tools/probes/ab/$L      # tfmpc/_hip.py loads this build; the product library is never touched
  echo -n "$L: "; python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['status_flagged_instances'])"
done; done
