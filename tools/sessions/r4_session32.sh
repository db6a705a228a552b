#!/bin/bash
# round 4, GPU visit 32: random programs -- one op per call vs range forms vs compiled in
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s32}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
( time timeout 900 python -m pytest tests/test_rowspec.py -m gpu -x -q -k "random_programs" > $O/pytest.log 2>&1 ) 2> $O/t.time; echo "tests rc=$?"; tail -15 $O/pytest.log | cut -c1-300; grep real $O/t.time
