#!/bin/bash
# PMC passes (one counter group per run, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
# do not fit one pass) around a short eager-mode bench run.   usage: tools/pmc.sh <tag> <bench args...>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/pmc_$TAG
# PMC_TRAFFIC_ONLY=1: the two traffic passes only
if [ "${PMC_TRAFFIC_ONLY:-0}" == "1" ]; then GROUPS_=("FETCH_SIZE" "WRITE_SIZE"); else GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"); fi
for grp in "${GROUPS_[@]}"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/$name -o x -- \
      python $R/bench.py --mode eager --protocol resident --steps 40 --warmup 5 --repeats 1 --no-cpu-baseline --no-extra "$@" > $R/gpurun_out/pmc_$TAG/$name.log 2>&1
done
find $R/gpurun_out/pmc_$TAG -name "*.csv" | head -20
