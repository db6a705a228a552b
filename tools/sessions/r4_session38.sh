#!/bin/bash
# round 4, GPU visit 38: nt as a launch-uniform branch (half the kernels and image entry points), outputs of non-final steps skipped in
# trajectory-less rollouts: tests, ablation, rollout rates
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s38}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_rowspec.py tests/test_f3_scenarios.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rollout+rowspec+f3 rc=$?"; tail -6 $O/pytest.log | cut -c1-300
timeout 600 python tools/rows_ablate.py > $O/rows_ablation.txt 2> $O/rows_ablation.err; echo "ablate rc=$?"; grep "full step\|COMPILED\|^simple\|fused kernel" $O/rows_ablation.txt
timeout 600 python tools/rows_rollout_rate.py > $O/rollout_rate.txt 2> $O/rollout_rate.err; echo "rate rc=$?"; cat $O/rollout_rate.txt; grep -v amdgpu.ids $O/rollout_rate.err | tail -5
