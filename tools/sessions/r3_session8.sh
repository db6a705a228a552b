#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s8}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_rollout.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
REPS=2 tools/ab_matrix.sh $TAG "tag:3:16384 spread:3:4096 spread:3:16384 simple_adversary:3:16384 simple_push:2:16384 simple:1:16384 spread:4:16384 spread:6:8192" nodual rewall ownall base
