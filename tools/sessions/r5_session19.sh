#!/bin/bash
# round 5, GPU visit 19: the bench line with its timeline (visit 18's run took 4 minutes instead of 72 s: where?)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s19}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench rc=$?"; grep real $O/bench_20.time
python - <<PY
import json
d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
print("value %.3f G" % (d["value"]/1e9), d["config"]["timeline_s"], "total before cpu baseline %.0f" % d["config"]["wall_s_since_start"])
PY
exit 0
