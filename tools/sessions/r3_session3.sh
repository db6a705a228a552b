#!/bin/bash
# round 3, GPU visit 3: parity suite (+ parity_r3.json), driver-style bench, kernel traces of every BASELINE config, PMC for C2, C4 x 3 processes
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s3}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_r3.json
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
cp $R/gpurun_out/parity_r3.json $O/ 2>/dev/null
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench20 rc=$?"; tail -3 $O/bench_20.time
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
    print("value %.3f G ms/step %.5f region %.2fs k_us %.3f frac %.3f" % (d["value"]/1e9, d["ms_per_step"], d["config"]["timed_region_s"], d["roofline"]["kernel_us_per_launch"], d["roofline"]["frac"]))
    for k,v in d.get("extra",{}).items():
        if k=="configs":
            for kk,vv in v.items(): print("  ", kk, "%.4g" % vv["value"], "k_us %.2f frac %.3f" % (vv["roofline"]["kernel_us_per_launch"], vv["roofline"]["frac"]), "| roll k_us %.2f frac_c %.3f" % (vv["fused_rollout"]["kernel_us_per_step"], vv["fused_rollout"]["frac_compulsory"]))
        elif k=="hbm_resident": print("  1M: %.4g k_us %.2f frac %.3f" % (v["value"], v["roofline"]["kernel_us_per_launch"], v["roofline"]["frac"]))
        elif k!="box": print("  ", k, "%.4g" % v["value"], v.get("kernel_us_per_step"))
except Exception as e: print("parse failed", e)
PY
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
trace rollout_tag_B16384 --scenario simple_tag --batch 16384 --steps 200 --mode fused
trace rollout_spread3_B4096 --batch 4096 --steps 200 --mode fused
for cfg in "spread3_B4096 --batch 4096" "tag_B16384 --scenario simple_tag --batch 16384"; do
  set -- $cfg; name=$1; shift
  tools/pmc.sh ${TAG}_$name "$@" > /dev/null 2>&1
  python profiles/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$name k_split > $O/pmc_$name.txt 2>>$O/err.log
  rm -rf $R/gpurun_out/pmc_${TAG}_$name        # (raw counter CSVs: tens of MB; the summary is what travels back)
done
for k in 1 2 3; do
  timeout 200 python bench.py --agents 64 --batch 4096 --steps 50 --warmup 10 --no-extra --no-cpu-baseline --region-ms 300 >> $O/c4_processes.jsonl 2>> $O/c4.err
done
python - <<PY
import json
for l in open("$O/c4_processes.jsonl"):
    d=json.loads(l); print("C4 process: k_us %.2f value %.4g frac %.3f | %s" % (d["roofline"]["kernel_us_per_launch"], d["value"], d["roofline"]["frac"], d["extra"]["box"].get("uuid")))
PY
ls $O
