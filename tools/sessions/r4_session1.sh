#!/bin/bash
# round 4, GPU visit 1: device-clock spans (C2 / C3 / C5 / 1M), the hardened N>1 bench paths, the ADVICE regression tests
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s1}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python tools/device_span.py C2 C3 C5 1M --out $O > $O/span.log 2> $O/span.err; echo "span rc=$?"; tail -40 $O/span.log; tail -5 $O/span.err
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q > $O/pytest_multirank.log 2>&1; echo "multirank rc=$?"; tail -8 $O/pytest_multirank.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_f3_scenarios.py -m gpu -x -q -k "rollout_in_between or movable_landmark or fast_path" > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 $O/pytest_new.log
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench20 rc=$?"; tail -3 $O/bench_20.time; tail -c 600 $O/bench_20.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.3f G ms/step %.5f k_us %.3f frac %.3f frac_timed_region %.3f floor %.2f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_us_per_launch"], r["frac"], r["frac_timed_region"], r["launch_floor_us"]))
    for kk,vv in d["extra"]["configs"].items(): print("  ", kk, "%.4g" % vv["value"], "ms/step %.5f k_us %.3f frac %.3f" % (vv["ms_per_step"], vv["roofline"]["kernel_us_per_launch"], vv["roofline"]["frac"]))
    v=d["extra"]["hbm_resident"]; print("  1M k_us %.2f frac %.3f" % (v["roofline"]["kernel_us_per_launch"], v["roofline"]["frac"]))
except Exception as e: print("parse failed", e)
PY
