#!/bin/bash
# round 5, GPU visit 17: the shared reward values' barrier under the delayed-wave image (and without it: the negative control)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s17}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_race.py tests/test_gpu_traced.py -m gpu -q --durations=4 -k "shared or traced" > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
exit 0
