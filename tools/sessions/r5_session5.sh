#!/bin/bash
# round 5, GPU visit 5: test_gpu_traced with durations (the suite went from 2 to 9 minutes in visit 4: where?)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s5}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_traced.py tests/test_refstyle.py -m gpu -q --durations=12 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -40 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
( time timeout 1200 python -m pytest tests -m gpu -q --durations=12 --deselect tests/test_gpu_traced.py --deselect tests/test_refstyle.py > $O/pytest2.log 2>&1 ) 2> $O/pytest2.time; echo "pytest rc=$?"; tail -25 $O/pytest2.log | cut -c1-300; grep real $O/pytest2.time
