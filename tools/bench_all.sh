#!/bin/bash
# One bench.py line per fused scenario (B = 65 536) + the other BASELINE shapes -> gpurun_out/<tag>/bench_<name>.json
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-all}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for sc in ${SCENARIOS:-simple simple_adversary simple_crypto simple_push simple_reference simple_speaker_listener simple_world_comm}; do
  timeout 300 python bench.py --scenario $sc --steps 200 --warmup 20 --cpu-seconds 2 > $O/bench_$sc.json 2> $O/bench_$sc.err
done
if [ -z "${ONLY_SCENARIOS:-}" ]; then
timeout 300 python bench.py --scenario simple_tag --batch 16384 --steps 200 --warmup 20 --cpu-seconds 2 > $O/bench_tag.json 2> $O/bench_tag.err
timeout 300 python bench.py --agents 64 --batch 4096 --steps 200 --warmup 20 --cpu-seconds 2 > $O/bench_n64.json 2> $O/bench_n64.err
timeout 300 python bench.py --batch 1048576 --steps 25 --warmup 5 --cpu-seconds 2 > $O/bench_1M.json 2> $O/bench_1M.err
timeout 300 python bench.py --steps 1000 --warmup 50 --cpu-seconds 2 > $O/bench_default_steps1000.json 2> $O/bench_default_steps1000.err
fi
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; fr=d.get("extra",{}).get("fused_rollout",{})
    print("%-44s value %.4g  kernel_us %.2f frac %.3f | fused us/step %s" % (sys.argv[1].split("/")[-1], d["value"], r["kernel_us_per_launch"], r["frac"], fr.get("kernel_us_per_step")))
except Exception as e: print(sys.argv[1], "failed", e)
PY
done
