#!/bin/bash
# the fill-shaped move generator: bit-exactness tests, then the headline with it and with the old generator, same box
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s45}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_abi.py tests/test_gpu_rollout.py -m gpu -x -q -k "random or philox or regenerated or moves or fused_rollout_equals_stepwise or abi" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for rep in 1 2; do
for v in base oldgen; do
  if [ $v == base ]; then unset MPE_HIP_LIB; else export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_$v.so; fi
  timeout 200 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 > $O/b_$v.json
  python -c "
import json; d=json.load(open('$O/b_$v.json')); print('$v', 'value %.4f G  us/step %.4f  k_us %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['kernel_us_per_launch']))" | tee -a $O/ab.log
done
done
unset MPE_HIP_LIB
