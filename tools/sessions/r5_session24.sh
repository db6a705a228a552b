#!/bin/bash
# round 5, GPU visit 24 (final code, with the reset riding on the block draw): the full set on the final code -- GPU suite (+ parity_r5.json), smoke(), the driver-style bench line, the one-GPU
# rehearsal of --gpus 2, the example (kernel traces / PMC of the unchanged step kernels: r5_session10.sh)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s24}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_r5.json
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -14 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
cp $R/gpurun_out/parity_r5.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -7
timeout 300 python examples/run_reference_style_file.py 2>&1 | grep -v amdgpu.ids | tail -4
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench rc=$?"; grep real $O/bench_20.time
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.3f G ms/step %.5f k_us %.3f frac %.3f frac_timed_region %.3f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_us_per_launch"], r["frac"], r["frac_timed_region"]))
    for kk,vv in d["extra"]["configs"].items(): print("  ", kk, "%.4g" % vv["value"], "k_us %.3f frac %.3f roll %.2f" % (vv["roofline"]["kernel_us_per_launch"], vv["roofline"]["frac"], vv["fused_rollout"]["kernel_us_per_step"]))
    v=d["extra"]["hbm_resident"]; print("  1M k_us %.2f frac %.3f" % (v["roofline"]["kernel_us_per_launch"], v["roofline"]["frac"]))
    print("  python_api %.3g" % d["extra"]["python_api"]["value"])
    u=d["extra"]["user_scenario"]
    for k in ("program","compiled","generic","compiled_fused_rollout"): print("  user_scenario", k, "%.4g" % u[k]["value"])
    u=d["extra"]["reference_style_file"]
    print("  reference_style_file roofline", {k: u["roofline"][k] for k in ("kernel_us_per_launch","algorithmic_bytes_per_env_step","frac")})
    for k in ("traced","traced_graph","traced_fused_rollout","host_path"): print("  reference_style_file", k, "%.4g" % u[k]["value"])
    print("  cpu_baseline", json.dumps(d["cpu_baseline"])[:200])
except Exception as e: print("parse failed", repr(e))
PY
timeout 300 python bench.py --gpus 2 --all-ranks-on-gpu0 --steps 20 --warmup 5 --no-extra 2>$O/bench_n2.err | tail -1 > $O/bench_n2.json; python -c "
import json; d=json.load(open('$O/bench_n2.json')); print('N=2 rehearsal: n_gpus', d['n_gpus'], 'value %.3g' % d['value'], d['config'].get('barrier_backend'))"
ls $O
exit 0
