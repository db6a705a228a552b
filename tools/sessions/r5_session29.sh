#!/bin/bash
# round 5, GPU visit 29 (after array comparisons / axis reductions / continue were predicated: tests/refstyle/mesh.py, the new trace of
# simple_crypto.py): the full GPU suite (+ parity_r5.json), smoke(), mesh.py traced vs its host path
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s29}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_r5.json
( time timeout 1200 python -m pytest tests -m gpu -q --durations=4 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -9 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
cp $R/gpurun_out/parity_r5.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python tools/refstyle_rate.py tests/refstyle/mesh.py 2>&1 | grep -v amdgpu.ids > $O/refstyle_rate_mesh.txt; tail -4 $O/refstyle_rate_mesh.txt | cut -c1-330
timeout 300 python tools/refstyle_rate.py --json simple_crypto 2>&1 | grep -v amdgpu.ids > $O/traced_vs_fused_crypto.txt; tail -4 $O/traced_vs_fused_crypto.txt | cut -c1-330
ls $O
exit 0
