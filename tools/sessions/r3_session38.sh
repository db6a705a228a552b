#!/bin/bash
# traffic counters (FETCH_SIZE / WRITE_SIZE passes) of the final code for the headline, the 1M leg and C4
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s38}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export PMC_TRAFFIC_ONLY=1
for cfg in "spread3_B65536 k_split" "spread3_B1M k_split --batch 1048576 --steps 10" "spread64_B4096 k_duo --agents 64 --batch 4096 --steps 10"; do
  set -- $cfg; name=$1; pat=$2; shift 2
  timeout 400 tools/pmc.sh ${TAG}_$name "$@" > /dev/null 2>&1
  python profiles/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$name $pat > $O/pmc_$name.txt 2>>$O/err.log
  rm -rf $R/gpurun_out/pmc_${TAG}_$name
  grep "traffic_bytes\|Kernel_Name" $O/pmc_$name.txt
done
