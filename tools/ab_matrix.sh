#!/bin/bash
# tools/ab_matrix.sh <out-tag> "<spec> <spec> ..." <variant> [<variant> ...]   (variant "base" = the normal build); REPS=2
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; SPECS=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    if [ $v == base ]; then unset MPE_HIP_LIB; else export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_$v.so; fi
    timeout 300 python tools/ab_kernels.py $SPECS 2>> $O/err.log | tee -a $O/matrix.txt
  done
done
unset MPE_HIP_LIB
