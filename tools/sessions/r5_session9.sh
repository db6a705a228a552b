#!/bin/bash
# round 5, GPU visit 9: done programs with a reset_world of their own (restart through reset_callback); the row-program suite
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s9}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_rowspec.py tests/test_gpu_traced.py tests/test_gpu_rollout.py -m gpu -q --durations=5 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
exit 0
