#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s7}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
rm -f $R/gpurun_out/parity_r3.json
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
cp $R/gpurun_out/parity_r3.json $O/ 2>/dev/null
tools/ab_matrix.sh $TAG "tag:3:16384 tag:3:32768 spread:3:4096 spread:3:16384 spread:3:32768 spread:3:65536 simple_adversary:3:16384 simple:1:65536 simple_push:2:16384" nodual dualall base
export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_clk.so
timeout 120 python tools/phase_clock.py simple_tag 16384 roll 2>>$O/err.log | tee -a $O/phase.txt
timeout 120 python tools/phase_clock.py simple_spread 65536 step 2>>$O/err.log | tee -a $O/phase.txt
unset MPE_HIP_LIB
for k in 1 2 3; do
  timeout 200 python bench.py --agents 64 --batch 4096 --steps 50 --warmup 10 --no-extra --no-cpu-baseline --region-ms 300 >> $O/c4_processes.jsonl 2>> $O/c4.err
done
python - <<PY
import json
for l in open("$O/c4_processes.jsonl"):
    d=json.loads(l); print("C4 process: k_us %.2f value %.4g frac %.3f | probe %s" % (d["roofline"]["kernel_us_per_launch"], d["value"], d["roofline"]["frac"], d["config"].get("placement_probe")))
PY
tail -3 $O/err.log $O/c4.err
