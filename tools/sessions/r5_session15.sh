#!/bin/bash
# round 5, GPU visit 15: shared reward sub-graphs computed once per world (traced_shared): the row-program / traced / rollout suites
# (RowDims and the LDS layout changed), team sizes N = 4 .. 16
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s15}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_traced.py tests/test_refstyle.py tests/test_rowspec.py tests/test_gpu_rollout.py tests/test_gpu_abi.py -m gpu -q > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -25 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
timeout 900 python tools/refstyle_rate.py --nav 4 --nav 6 --nav 8 --nav 10 --nav 12 --nav 16 > $O/team_sizes.txt 2> $O/team_sizes.err; echo "rc=$?"; cat $O/team_sizes.txt; tail -5 $O/team_sizes.err | grep -v amdgpu
exit 0
