#!/bin/bash
# round 5, GPU visit 26: the driver-style bench line on another box (final code)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s26}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench rc=$?"; grep real $O/bench_20.time
python - <<PY
import json
d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value %.3f G ms/step %.5f k_us %.3f frac %.3f frac_timed_region %.3f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_us_per_launch"], r["frac"], r["frac_timed_region"]), d["config"]["timeline_s"])
for kk,vv in d["extra"]["configs"].items(): print("  ", kk, "%.4g" % vv["value"], "k_us %.3f frac %.3f" % (vv["roofline"]["kernel_us_per_launch"], vv["roofline"]["frac"]))
u=d["extra"]["reference_style_file"]; print("  traced", "%.4g" % u["traced"]["value"], "graph %.4g" % u["traced_graph"]["value"], "frac %.3f" % u["roofline"]["frac"])
PY
exit 0
