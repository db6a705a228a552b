#!/bin/bash
# round 4, GPU visit 23: the rest of the rowspec tests after the sticky-error fix; what done_callback + auto_reset costs when nothing finishes
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s23}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 900 python -m pytest tests/test_rowspec.py tests/test_gpu_abi.py -m gpu -x -q > $O/pytest_rowspec.log 2>&1; echo "rowspec+abi rc=$?"; tail -5 $O/pytest_rowspec.log | cut -c1-300
timeout 600 python tools/finish_cost.py > $O/finish_cost.txt 2> $O/finish_cost.err; echo "finish_cost rc=$?"; cat $O/finish_cost.txt; grep -v amdgpu.ids $O/finish_cost.err | tail -5
