#!/bin/bash
# Round profile set: for each bench workload a rocprofv3 --kernel-trace --stats pass of the bench command
# and the PMC passes (tools/pmc.sh).  usage: tools/gpu_profile.sh <tag>     -> gpurun_out/<tag>/...
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-prof}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
run() {  # name, bench args...
  local name=$1; shift
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o x -- \
      python $R/bench.py --no-cpu-baseline --no-extra --repeats 2 "$@" > $O/$name.bench.json 2> $O/$name.err
  cp $(find $O/$name -name "x_kernel_stats.csv" | head -1) $O/$name.kernel_stats.csv
  cd $R
  tools/pmc.sh ${TAG}_$name "$@" > /dev/null 2>&1
}
run spread3_B65536
run spread64_B4096 --agents 64 --batch 4096 --steps 200
run tag_B16384 --scenario simple_tag --batch 16384
ls $O
