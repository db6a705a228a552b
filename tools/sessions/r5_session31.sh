#!/bin/bash
# round 5, GPU visit 31 (host resets with picks; array constructors through the file's numpy proxy): the full GPU suite
# (+ parity_r5.json), smoke(), scatter.py traced vs its host path, the driver-style bench line
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s31}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_r5.json
( time timeout 1200 python -m pytest tests -m gpu -q --durations=4 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -9 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
cp $R/gpurun_out/parity_r5.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python tools/refstyle_rate.py tests/refstyle/scatter.py 2>&1 | grep -v amdgpu.ids > $O/refstyle_rate_scatter.txt; tail -4 $O/refstyle_rate_scatter.txt | cut -c1-330
( time timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench rc=$?"; grep real $O/bench_20.time
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.3f G ms/step %.5f k_us %.3f frac %.3f frac_timed_region %.3f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_us_per_launch"], r["frac"], r["frac_timed_region"]))
    u=d["extra"]["reference_style_file"]
    for k in ("traced","traced_graph","traced_fused_rollout","host_path"): print("  reference_style_file", k, "%.4g" % u[k]["value"])
except Exception as e: print("parse failed", repr(e))
PY
ls $O
exit 0
