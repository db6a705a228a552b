#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s32}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "fast_path or partial or graphed or race or abi or multirank" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
print("value %.3f G k_us %.3f frac %.3f" % (d["value"]/1e9, d["roofline"]["kernel_us_per_launch"], d["roofline"]["frac"]))
for k in ("moves_resident","int_action_ids","python_api","host_buffers"):
    print("  ", k, "%.4g" % d["extra"][k]["value"], {a:b for a,b in d["extra"][k].items() if a not in ("value","note","what")})
PY
