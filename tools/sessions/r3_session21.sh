#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s21}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do
for v in base sc1 sc1w2 sc1g2w2 sc1w1; do
  if [ $v == base ]; then unset MPE_HIP_LIB; else export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_$v.so; fi
  timeout 200 python tools/c4_placement.py 10 brief 2>&1 | grep -v amdgpu.ids >> $O/occ.log
done
done
cat $O/occ.log
timeout 300 python tools/write_pattern.py 10 2>&1 | grep -v amdgpu.ids > $O/wp.log; cat $O/wp.log
