#!/bin/bash
# why are C2 / C3 ~15 % slower inside the default bench run than in a process of their own?
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s9}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
one() { timeout 200 python bench.py --scenario simple_tag --batch 16384 --steps 200 --no-extra --no-cpu-baseline --region-ms 200 2>>$O/err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 tag standalone: k_us %.3f ms/step %.5f  sclk %s power %s' % (d['roofline']['kernel_us_per_launch'], d['ms_per_step'], d['extra']['box'].get('rocm_smi',{}).get('card0/sclk clock speed:'), d['extra']['box'].get('rocm_smi',{}).get('card0/Current Socket Graphics Package Power (W)')))"; }
one cold1; one cold2
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2>>$O/err.log
python - <<PY
import json
d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
print("in-bench: headline k_us %.3f" % d["roofline"]["kernel_us_per_launch"])
for kk,vv in d["extra"]["configs"].items(): print("  ", kk, "k_us %.2f | roll %.2f" % (vv["roofline"]["kernel_us_per_launch"], vv["fused_rollout"]["kernel_us_per_step"]))
PY
one hot1; one hot2
REPS=1 tools/ab_matrix.sh $TAG "tag:3:16384 spread:3:4096 spread:3:65536" base
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
