#!/bin/bash
# round 5, GPU visit 6: the traced path with predicated control flow (new graphs, new images): its GPU tests, the fixtures' rates
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s6}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_traced.py tests/test_refstyle.py -m gpu -q --durations=5 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -14 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
timeout 600 python tools/refstyle_rate.py > $O/refstyle_rate.txt 2> $O/refstyle_rate.err; echo "refstyle_rate rc=$?"; cat $O/refstyle_rate.txt; tail -8 $O/refstyle_rate.err | grep -v amdgpu
