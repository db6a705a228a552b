#!/bin/bash
# round 5, GPU visit 4: the full GPU suite with the traced path, smoke() (incl. the traced file), rates of the traced fixtures,
# the driver-style bench line with extra.reference_style_file
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s4}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6
timeout 600 python tools/refstyle_rate.py > $O/refstyle_rate.txt 2> $O/refstyle_rate.err; echo "refstyle_rate rc=$?"; cat $O/refstyle_rate.txt; tail -8 $O/refstyle_rate.err | grep -v amdgpu
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench rc=$?"; grep real $O/bench_20.time; tail -5 $O/bench_20.err | grep -v amdgpu
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.3f G ms/step %.5f k_us %.3f frac %.3f frac_timed_region %.3f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_us_per_launch"], r["frac"], r["frac_timed_region"]))
    u=d["extra"]["reference_style_file"]
    print("  build_s %.2f trace %s" % (u["build_s"], u["trace"]))
    for k in ("traced","traced_graph","traced_fused_rollout","host_path"): print("  reference_style_file", k, "%.4g" % u[k]["value"], {kk: vv for kk, vv in u[k].items() if kk.startswith(("us_","ms_"))})
    print("  traced_over_host_path %.0f" % u["traced_over_host_path"])
    u=d["extra"]["user_scenario"]
    for k in ("program","compiled","generic","compiled_fused_rollout"): print("  user_scenario", k, "%.4g" % u[k]["value"])
except Exception as e: print("parse failed", repr(e))
PY
