#!/bin/bash
# round 5, GPU visit 32 (random numbers kept outside the state as per-world parameters in the pick slots: survey.py's rally point):
# the full GPU suite (+ parity_r5.json), smoke(), survey.py traced vs its host path (graph protocol WITH device restarts)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s32}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_r5.json
( time timeout 1200 python -m pytest tests -m gpu -q --durations=4 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -9 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
cp $R/gpurun_out/parity_r5.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python tools/refstyle_rate.py tests/refstyle/survey.py 2>&1 | grep -v amdgpu.ids > $O/refstyle_rate_survey.txt; tail -4 $O/refstyle_rate_survey.txt | cut -c1-330
ls $O
exit 0
