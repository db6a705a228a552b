#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s12}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2 3; do for m in single double triple; do timeout 120 python tools/c4_first.py $m 2>>$O/err.log | tee -a $O/first.txt; done; done
tail -2 $O/err.log
