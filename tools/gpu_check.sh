#!/bin/bash
# One GPU-box visit: parity tests, the bench lines, the N=64 probe.   usage: tools/gpu_check.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-chk}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $O/b_spread3.json 2> $O/b_spread3.err; tail -c 600 $O/b_spread3.json
timeout 300 python bench.py --no-cpu-baseline --agents 64 --batch 4096 --steps 200 > $O/b_n64.json 2> $O/b_n64.err; tail -c 700 $O/b_n64.json
timeout 300 python bench.py --no-cpu-baseline --scenario simple_tag --batch 16384 > $O/b_tag.json 2> $O/b_tag.err; tail -c 500 $O/b_tag.json
timeout 300 python tools/probe_wide.py 64 4096 > $O/probe_wide.log 2>&1; cat $O/probe_wide.log
