#!/bin/bash
# round 4, GPU visit 10: VMM order experiment (consecutive vs shuffled physical chunks), the second and last VMM session
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s10}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 900 python tools/vmm_c4_order.py > $O/vmm_order.txt 2> $O/vmm_order.err; echo "vmm order rc=$?"; cat $O/vmm_order.txt | cut -c1-200; tail -5 $O/vmm_order.err
