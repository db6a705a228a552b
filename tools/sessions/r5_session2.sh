#!/bin/bash
# round 5, GPU visit 2: first run of the traced path of reference-style files (symtrace.py) + the GPU suite with the round-4 review fixes
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s2}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
timeout 600 python tools/refstyle_rate.py > $O/refstyle_rate.txt 2> $O/refstyle_rate.err; echo "refstyle_rate rc=$?"; cat $O/refstyle_rate.txt; tail -5 $O/refstyle_rate.err
( time timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -15 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
