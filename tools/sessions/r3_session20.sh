#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s20}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in 1 2; do timeout 300 python tools/write_pattern.py 8 >> $O/wp.log 2>&1; echo "--" >> $O/wp.log; done
cat $O/wp.log
