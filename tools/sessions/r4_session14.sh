#!/bin/bash
# round 4, GPU visit 14: phase clocks of k_rows (where a row-program step's cycles go)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s14}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
for s in simple_spread simple_tag simple_world_comm; do
  MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_rowsclock.so timeout 200 python tools/rows_clock.py $s >> $O/rows_clock.txt 2>> $O/rows_clock.err
done
cat $O/rows_clock.txt; tail -3 $O/rows_clock.err
