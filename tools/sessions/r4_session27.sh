#!/bin/bash
# round 4, GPU visit 27: what the in-launch episode end (second pass) costs the plain program step -- product vs the noep2 A/B build,
# interpreted and compiled in; the rowspec / f3 tests with the episode numbering fixed
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s27}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 900 python -m pytest tests/test_rowspec.py tests/test_f3_scenarios.py -m gpu -x -q > $O/pytest_rowspec.log 2>&1; echo "rowspec+f3 rc=$?"; tail -4 $O/pytest_rowspec.log | cut -c1-300
for k in 1 2; do
  echo "== product"; timeout 600 python tools/rows_ablate.py 2> $O/a.err | grep "full step\|COMPILED\|^simple"
  echo "== noep2"; MPE_ROWS_IMAGE_FLAGS="-DMPE_ROWS_NO_EPISODE2" MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_noep2.so timeout 600 python tools/rows_ablate.py 2> $O/b.err | grep "full step\|COMPILED\|^simple"
done
