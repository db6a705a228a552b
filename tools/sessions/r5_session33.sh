#!/bin/bash
# round 5, GPU visit 33 (float32 rows in mesh.py, helper modules, early return + assignments): the full GPU suite (+ parity_r5.json), smoke()
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s33}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_r5.json
( time timeout 900 python -m pytest tests -m gpu -q --durations=4 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -9 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
cp $R/gpurun_out/parity_r5.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
ls $O
exit 0
