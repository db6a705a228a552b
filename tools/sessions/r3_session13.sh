#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s13}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_abi.py tests/test_gpu_rollout.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2; do for v in actdw base; do
  if [ $v == base ]; then unset MPE_HIP_LIB; else export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_$v.so; fi
  timeout 200 python bench.py --steps 200 --no-extra --no-cpu-baseline --region-ms 400 --repeats 3 2>>$O/err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v headline value %.4f G  ms/step %.5f  k_us %.3f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_us_per_launch']))" | tee -a $O/act.txt
  timeout 200 python - <<PY | tee -a $O/act.txt
import torch, ctypes as C
import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _abi
L=_abi.lib(); A,B,T=3,65536,25
t=torch.empty((T,A,B,5),device="cuda"); st=_abi.raw_stream(t.device)
for _ in range(5): L.mpe_random_actions_block(t.data_ptr(), None, A, B, 1, 0, T, 0, st)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(100): L.mpe_random_actions_block(t.data_ptr(), None, A, B, 1, 25*k, T, 0, st)
e1.record(); torch.cuda.synchronize()
us=e0.elapsed_time(e1)*10
print("$v  mpe_random_actions_block(25 steps, A=3, B=65536): %.2f us per launch = %.2f TB/s" % (us, t.numel()*4/us/1e6))
PY
done; done
unset MPE_HIP_LIB
