#!/bin/bash
# round 5, GPU visit 13: the generated code with min(sqrt, sqrt) as sqrt(min) -- traced GPU tests, team sizes, the nine files beside their
# hand-fused kernels again
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s13}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_traced.py tests/test_refstyle.py -m gpu -q > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
timeout 900 python tools/refstyle_rate.py --nav 4 --nav 6 --nav 8 --nav 10 --nav 12 > $O/team_sizes.txt 2> $O/team_sizes.err; echo "rc=$?"; cat $O/team_sizes.txt
timeout 900 python tools/refstyle_rate.py --json simple_spread --json simple_tag --json simple_adversary --json simple_push --json simple_reference --json simple_speaker_listener --json simple_crypto --json simple_world_comm --json simple tests/refstyle/convoy.py > $O/traced_vs_fused.txt 2> $O/traced_vs_fused.err; echo "traced_vs_fused rc=$?"; cat $O/traced_vs_fused.txt
exit 0
