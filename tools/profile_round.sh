#!/bin/bash
# Round-end measurements on the GPU box, into gpurun_out/final/ (copy what is to be judged into profiles/):
#   cfg5 PMC passes (tools/pmc_cfg5.sh), rocprofv3 --kernel-trace --stats of bench.py (headline kernel) and of
#   tools/all_kernels_once.py (every kernel family), and the full bench.py line.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_round.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/final
mkdir -p $O
bash $ROOT/tools/pmc_cfg5.sh final > $O/cfg5_pmc.json 2> $O/cfg5_pmc.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/all_stats -- python3 $ROOT/tools/all_kernels_once.py > $O/all_kernels.log 2>&1
cd $ROOT && python3 bench.py > $O/bench.json 2> $O/bench.err
find $O -name "*_kernel_stats.csv" | head
tail -c 600 $O/bench.json
