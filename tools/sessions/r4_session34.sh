#!/bin/bash
# round 4, GPU visit 34: RandomRollout (enqueue / capture) on row-program envs; the bench line with the graph-replayed user scenario
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s34}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 900 python -m pytest tests/test_gpu_rollout.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rollout tests rc=$?"; tail -12 $O/pytest.log | cut -c1-300
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench rc=$?"; grep real $O/bench_20.time; tail -c 600 $O/bench_20.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.3f G ms/step %.5f k_us %.3f frac %.3f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_us_per_launch"], r["frac"]))
    u=d["extra"]["user_scenario"]
    for k in ("program","compiled","generic"): print("  user_scenario", k, "%.4g" % u[k]["value"], "%.2f us" % u[k]["us_per_step"], u[k]["path"])
    for k in ("program_graph","compiled_graph"): print("  user_scenario", k, "%.4g" % u[k]["value"], "%.3f us" % (u[k]["ms_per_step"]*1e3), u[k]["path"])
except Exception as e: print("parse failed", e)
PY
