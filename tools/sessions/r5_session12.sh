#!/bin/bash
# round 5, GPU visit 12: how far straight-line traced code carries -- N x N cooperative navigation as a reference-style file, N = 4 .. 12,
# beside the built-in simple_spread of that size
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s12}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
timeout 900 python tools/refstyle_rate.py --nav 4 --nav 6 --nav 8 --nav 10 --nav 12 > $O/team_sizes.txt 2> $O/team_sizes.err; echo "rc=$?"; cat $O/team_sizes.txt; tail -5 $O/team_sizes.err | grep -v amdgpu
exit 0
