#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s30}; cd $R
REPS=4 tools/ab_matrix.sh $TAG "spread:64:4096" base nt
