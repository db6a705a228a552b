#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s40}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in 1 2 3; do timeout 200 python tools/c4_placement.py 8 brief 3 1048576 2>&1 | grep -v amdgpu.ids | tee -a $O/p1m.log; done
