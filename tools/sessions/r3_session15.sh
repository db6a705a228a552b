#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s15}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
REPS=2 tools/ab_matrix.sh $TAG "tag:3:16384 tag:3:65536 spread:3:65536 spread:3:4096 simple_world_comm:6:65536 spread:6:16384" nosl base
