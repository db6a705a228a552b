#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s11}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for k in 1 2 3 4 5 6; do
  timeout 200 python bench.py --agents 64 --batch 4096 --steps 50 --warmup 10 --no-extra --no-cpu-baseline --region-ms 300 >> $O/c4_processes.jsonl 2>> $O/c4.err
done
python - <<PY
import json
for l in open("$O/c4_processes.jsonl"):
    d=json.loads(l); print("C4 process: k_us %.2f value %.4g frac %.3f | probe %s" % (d["roofline"]["kernel_us_per_launch"], d["value"], d["roofline"]["frac"], d["config"].get("placement_probe")))
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "spread64 or large or N64 or two_waves" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
