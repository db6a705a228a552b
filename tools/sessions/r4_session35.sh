#!/bin/bash
# round 4, GPU visit 35: mpe_rollout_rows (fused T-step rollouts of row-program envs): tests, rates
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s35}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_rowspec.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rollout+rowspec rc=$?"; tail -12 $O/pytest.log | cut -c1-300
timeout 600 python tools/rows_rollout_rate.py > $O/rollout_rate.txt 2> $O/rollout_rate.err; echo "rate rc=$?"; cat $O/rollout_rate.txt; grep -v amdgpu.ids $O/rollout_rate.err | tail -5
