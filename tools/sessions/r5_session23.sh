#!/bin/bash
# round 5, GPU visit 23: reset_world and the episode's block of moves in one launch (mpe_reset_random_actions_block): the rollout suite,
# the bench line A/B (two launches vs one)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s23}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_abi.py tests/test_gpu_multirank.py -m gpu -q > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_fused_$i.json 2> $O/bench.err
MPE_NO_FUSED_RESET_DRAW=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_two_$i.json 2>> $O/bench.err
done
python - <<PY
import json
for k in ("fused_1","two_1","fused_2","two_2"):
    d=json.loads(open("$O/bench_%s.json" % k).read().strip().splitlines()[-1]); r=d["roofline"]
    print(k, "value %.3f G  us/step %.3f  frac_timed_region %.3f  k_us %.3f" % (d["value"]/1e9, d["ms_per_step"]*1e3, r["frac_timed_region"], r["kernel_us_per_launch"]))
PY
exit 0
