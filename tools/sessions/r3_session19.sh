#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s19}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in 1 2 3; do timeout 300 python tools/c4_stride.py 8 >> $O/stride.log 2>&1; echo "--" >> $O/stride.log; done
cat $O/stride.log
