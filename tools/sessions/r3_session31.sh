#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s31}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "partial_fusion" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -15 $O/tests.log
timeout 300 python tools/partial_obs_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/rate.log
