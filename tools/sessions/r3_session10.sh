#!/bin/bash
# C4: does the way the two waves of a world share the rows (split point / interleaved) change which buffers are "fast"?
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s10}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do
for v in base s12 s34 s78 s11 il; do
  if [ $v == base ]; then unset MPE_HIP_LIB; else export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_$v.so; fi
  timeout 200 python tools/c4_placement.py 8 brief 2>>$O/err.log | tee -a $O/split.txt
done
done
unset MPE_HIP_LIB
tail -3 $O/err.log
