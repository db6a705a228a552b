#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s5}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
rm -f $R/gpurun_out/parity_r3.json
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
cp $R/gpurun_out/parity_r3.json $O/ 2>/dev/null
for k in 1 2 3; do timeout 200 python tools/c4_placement.py 6 2>>$O/err.log | tee -a $O/placement_$k.txt; done
