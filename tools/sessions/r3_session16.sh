#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s16}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
REPS=2 tools/ab_matrix.sh $TAG "simple_speaker_listener:2:65536 simple_reference:2:65536 simple_adversary:3:65536 simple_push:2:65536 simple_crypto:3:65536 simple_world_comm:6:65536 tag:3:16384 spread:3:65536" scr base
