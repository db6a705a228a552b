#!/bin/bash
# round 5, GPU visit 10: the full set on the final code -- GPU suite (+ parity_r5.json), smoke(), the driver-style bench line, the one-GPU
# rehearsal of --gpus 2, kernel-trace summaries of every BASELINE config (r5 names), the traffic passes of the headline, the example
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s10}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
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
trace() {  # name, bench args...
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o x -- \
      python $R/bench.py --no-cpu-baseline --no-extra --repeats 2 --region-ms 40 "$@" > $O/trace_$name.bench.json 2> $O/trace_$name.err)
  local kt=$(find $O/trace_$name -name "x_kernel_trace.csv" | head -1)
  python tools/trace_summary.py $kt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra --repeats 2 --region-ms 40 $*" > $O/${name}_kernel_trace_summary.txt
  cp $(find $O/trace_$name -name "x_kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $O/trace_$name
  python -c "
import json; d=json.loads(open('$O/trace_$name.bench.json').read().strip().splitlines()[-1]); print('$name under the profiler: k_us %.3f ms/step %.5f' % (d['roofline']['kernel_us_per_launch'], d['ms_per_step']))"
  grep "^# period\|^# duration" $O/${name}_kernel_trace_summary.txt
}
trace spread3_B65536 --steps 200
trace spread3_B4096 --batch 4096 --steps 200
trace tag_B16384 --scenario simple_tag --batch 16384 --steps 200
trace spread64_B4096 --agents 64 --batch 4096 --steps 50 --warmup 10
trace spread3_B1M --batch 1048576 --steps 25 --warmup 5
export PMC_TRAFFIC_ONLY=1
for cfg in "spread3_B65536 k_split"; do
  set -- $cfg; name=$1; pat=$2; shift 2
  timeout 400 tools/pmc.sh ${TAG}_$name "$@" > /dev/null 2>&1
  python profiles/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$name $pat > $O/pmc_$name.txt 2>>$O/err.log
  rm -rf $R/gpurun_out/pmc_${TAG}_$name
  grep "traffic_bytes\|Kernel_Name" $O/pmc_$name.txt | cut -c1-200
done
ls $O
exit 0
