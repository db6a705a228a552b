#!/bin/bash
# round 5, GPU visit 25: A/B -- the block of moves written with nontemporal stores (libmpe_hip_ab_actnt.so) vs ordinary stores
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s25}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_base_$i.json 2> $O/bench.err
MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_actnt.so timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_nt_$i.json 2>> $O/bench.err
done
python - <<PY
import json
for k in ("base_1","nt_1","base_2","nt_2"):
    d=json.loads(open("$O/bench_%s.json" % k).read().strip().splitlines()[-1]); r=d["roofline"]
    print(k, "value %.3f G  us/step %.3f  frac_timed_region %.3f  k_us %.3f" % (d["value"]/1e9, d["ms_per_step"]*1e3, r["frac_timed_region"], r["kernel_us_per_launch"]))
PY
exit 0
