#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s17}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
REPS=2 tools/ab_matrix.sh $TAG "simple:1:65536 spread:3:65536 tag:3:65536 simple_adversary:3:65536 simple_push:2:65536 simple_speaker_listener:2:65536 simple_reference:2:65536 simple_crypto:3:65536 simple_world_comm:6:65536 spread:16:16384 spread:64:4096" r2 base
