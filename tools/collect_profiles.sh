#!/bin/bash
# Copies what is to be judged from gpurun_out/pmc_<round>* (tools/profile_r05.sh) into profiles/ under the round's names.
#   tools/collect_profiles.sh r05
R=${1:-r05}
cd "$(dirname "$0")/.."
cp gpurun_out/pmc_$R/summary.json profiles/${R}_mfma16x8_pmc_summary.json
cp $(ls -t gpurun_out/pmc_$R/stats/*/*_kernel_stats.csv | head -1) profiles/${R}_mfma16x8_kernel_stats.csv
grep "^{\"summary\"" gpurun_out/pmc_$R/stats.log | tail -1 > profiles/${R}_mfma16x8_bench_under_rocprof.json
for tag in ilqr_api ilqr_api_full box box_stable cfg5 cfg5_hvac cfg5_reservoir small_env hvac6 res4 large_tile lqr32 literal_dims cfg4 cfg4_one_launch cfg2; do
  d=gpurun_out/pmc_${R}_$tag
  [ -f $d/summary.json ] || continue
  cp $d/summary.json profiles/${R}_${tag}_pmc.json
  cp $(ls -t $d/stats/*/*_kernel_stats.csv | head -1) profiles/${R}_${tag}_kernel_stats.csv
done
ls -la profiles | grep "${R}_" | awk '{print $5, $9}'
