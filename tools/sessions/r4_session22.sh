#!/bin/bash
# round 4, GPU visit 22: compiled row programs -- policy tests (cached image attaches itself; hipcc on the box for a new one), the
# full GPU suite, the bench line with the compiled user-scenario leg
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s22}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
( time timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -8 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench rc=$?"; grep real $O/bench_20.time; tail -c 400 $O/bench_20.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.3f G ms/step %.5f k_us %.3f frac %.3f frac_timed_region %.3f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_us_per_launch"], r["frac"], r["frac_timed_region"]))
    u=d["extra"]["user_scenario"]
    for k in ("program","compiled","generic"): print("  user_scenario", k, "%.4g" % u[k]["value"], "%.2f us" % u[k]["us_per_step"], u[k]["path"])
    print("  compile_program_s %.2f  compiled/generic %.0f" % (u["compile_program_s"], u["compiled_over_generic"]))
except Exception as e: print("parse failed", e)
PY
