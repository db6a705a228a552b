#!/bin/bash
# round 5, GPU visit 27: the traced / refstyle / row-program suites after the last tracer changes (math wrappers), smoke
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s27}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
( time timeout 900 python -m pytest tests/test_gpu_traced.py tests/test_refstyle.py tests/test_rowspec.py tests/test_gpu_race.py -m gpu -q -k "not delayed_wave_build" > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
exit 0
