#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s43}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_abi.py -m gpu -x -q -k "bit_identical or abi" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
