#!/bin/bash
# round 4, GPU visit 24: mpe_episode_finish launched one wave per 64 worlds: the done_callback tests, its cost again
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s24}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 900 python -m pytest tests/test_rowspec.py tests/test_f3_scenarios.py tests/test_gpu_parity.py -m gpu -x -q -k "done or auto_reset or episode or finish or compiled" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -5 $O/pytest.log | cut -c1-300
timeout 600 python tools/finish_cost.py > $O/finish_cost.txt 2> $O/finish_cost.err; echo "finish_cost rc=$?"; cat $O/finish_cost.txt; grep -v amdgpu.ids $O/finish_cost.err | tail -5
