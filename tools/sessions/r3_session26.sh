#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s26}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 300 python tools/write_pattern.py 10 2>&1 | grep -v amdgpu.ids > $O/wp.log; cat $O/wp.log
