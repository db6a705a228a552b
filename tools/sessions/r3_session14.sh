#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s14}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do
for v in base rot1 rot4 rot32 rot256; do
  if [ $v == base ]; then unset MPE_HIP_LIB; else export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_$v.so; fi
  timeout 200 python tools/c4_placement.py 6 brief 2>>$O/err.log | grep "buffers\|output set" | tee -a $O/rot.txt
done
done
export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_rot4.so
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_waves or spread64 or large_worlds" 2>&1 | tail -2
unset MPE_HIP_LIB
tail -2 $O/err.log
