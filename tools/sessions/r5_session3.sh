#!/bin/bash
# round 5, GPU visit 3: the traced path's GPU tests (nine committed traces vs reference goldens, fixtures vs goldens and host path,
# episode ends, rollouts), the row-program / refstyle suites, rates of the traced fixtures
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s3}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_traced.py tests/test_refstyle.py tests/test_rowspec.py -m gpu -q > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -40 $O/pytest.log | cut -c1-400; grep real $O/pytest.time
timeout 600 python tools/refstyle_rate.py > $O/refstyle_rate.txt 2> $O/refstyle_rate.err; echo "refstyle_rate rc=$?"; cat $O/refstyle_rate.txt; tail -8 $O/refstyle_rate.err
