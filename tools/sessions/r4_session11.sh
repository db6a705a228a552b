#!/bin/bash
# round 4, GPU visit 11: where a row-program step's time goes (programs with parts removed)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s11}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 600 python tools/rows_ablate.py > $O/rows_ablation.txt 2> $O/rows_ablation.err; echo "ablate rc=$?"; cat $O/rows_ablation.txt; tail -5 $O/rows_ablation.err
