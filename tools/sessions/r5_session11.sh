#!/bin/bash
# round 5, GPU visit 11: per-entity reset boxes for row programs (mpe_reset_rows, in-kernel restarts, World.reset_boxes) -- the full
# GPU suite (k_rows' reset sites and RowTables changed), smoke, herd's rates with in-launch episodes
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s11}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python tools/refstyle_rate.py tests/refstyle/herd.py tests/refstyle/convoy.py > $O/refstyle_rate.txt 2> $O/refstyle_rate.err; echo "refstyle_rate rc=$?"; cat $O/refstyle_rate.txt
exit 0
