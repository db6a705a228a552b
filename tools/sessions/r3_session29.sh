#!/bin/bash
# k_duo rows at agent scope: the wide-kernel tests, then C4 through bench.py in 3 processes
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s29}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rollout.py tests/test_gpu_race.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
for i in 1 2 3; do timeout 300 python bench.py --agents 64 --batch 4096 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c4_$i.json
python - $O/c4_$i.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("C4 process: k_us %.2f frac %.3f value %.1f M | probe %s" % (r["kernel_us_per_launch"], r["frac"], d["value"]/1e6, d.get("extra",{}).get("placement_probe") or d["config"].get("placement_probe")))
P
done
