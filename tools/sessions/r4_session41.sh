#!/bin/bash
# round 4, GPU visit 41: the step as a lambda, rollouts whose episodes end by the done programs (mpe_rollout_rows_episode): tests
# (rollout, rowspec, f3, race), ablation, rollout rates
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s41}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_rowspec.py tests/test_f3_scenarios.py tests/test_gpu_race.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -8 $O/pytest.log | cut -c1-300
timeout 600 python tools/rows_ablate.py > $O/rows_ablation.txt 2> $O/rows_ablation.err; echo "ablate rc=$?"; grep "full step\|COMPILED\|^simple\|fused kernel" $O/rows_ablation.txt
timeout 600 python tools/rows_rollout_rate.py > $O/rollout_rate.txt 2> $O/rollout_rate.err; echo "rate rc=$?"; cat $O/rollout_rate.txt; grep -v amdgpu.ids $O/rollout_rate.err | tail -5
