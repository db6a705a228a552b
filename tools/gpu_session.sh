#!/bin/bash
# One GPU-box visit of round 2: parity tests, the driver-style bench lines, kernel-trace + PMC profiles.
#   usage: tools/gpu_session.sh <tag> [steps...]      steps: test bench bench1000 trace pmc1m pmc64k pmctag pmcn64 n64 tag
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-s}; shift
STEPS=${*:-test bench}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
has() { [[ " $STEPS " == *" $1 "* ]]; }
trace() {  # name, bench args...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o x -- \
      python $R/bench.py --no-cpu-baseline --no-extra --repeats 2 "$@" > $O/trace_$name.bench.json 2> $O/trace_$name.err)
  cp $(find $O/trace_$name -name "x_kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $O/trace_$name
  head -5 $O/${name}_kernel_stats.csv
}
if has test; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
  tail -15 $O/pytest.log
fi
if has bench; then
  timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?"; tail -c 1500 $O/bench_20.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
    print("value %.3f G  ms/step %.5f  R=%s  kernel_us %.3f frac %.3f floor %.2f" % (d["value"]/1e9, d["ms_per_step"], d["config"]["graph_replays_in_timed_region"], d["roofline"]["kernel_us_per_launch"], d["roofline"]["frac"], d["roofline"]["launch_floor_us"]))
    for k,v in d.get("extra",{}).items():
        if k=="configs":
            for kk,vv in v.items(): print("  ", kk, "%.4g steps/s" % vv["value"], "k_us %.2f frac %.3f" % (vv["roofline"]["kernel_us_per_launch"], vv["roofline"]["frac"]), "| rollout k_us %.2f frac_c %.3f" % (vv["fused_rollout"]["kernel_us_per_step"], vv["fused_rollout"]["frac_compulsory"]))
        elif k=="hbm_resident": print("  1M: %.4g steps/s k_us %.2f frac %.3f" % (v["value"], v["roofline"]["kernel_us_per_launch"], v["roofline"]["frac"]))
        else: print("  ", k, "%.4g" % v["value"], v.get("kernel_us_per_step"))
    print("  cpu", d.get("cpu_baseline",{}).get("value"))
except Exception as e: print("parse failed", e)
PY
fi
if has bench1000; then
  timeout 600 python bench.py --steps 1000 --warmup 50 --no-extra --no-cpu-baseline > $O/bench_1000.json 2> $O/bench_1000.err; echo "bench1000 rc=$?"
  python -c "
import json; d=json.loads(open('$O/bench_1000.json').read().strip().splitlines()[-1]); print('steps1000: value %.3f G ms/step %.5f' % (d['value']/1e9, d['ms_per_step']))"
fi
has trace && trace spread3_B65536 --steps 200
has trace1m && trace spread3_B1M --batch 1048576 --steps 50 --warmup 5
has tracetag && trace tag_B16384 --scenario simple_tag --batch 16384 --steps 200
has tracen64 && trace spread64_B4096 --agents 64 --batch 4096 --steps 100 --warmup 10
has pmc1m && tools/pmc.sh ${TAG}_spread3_B1M --batch 1048576 > /dev/null 2>&1
has pmc64k && tools/pmc.sh ${TAG}_spread3_B65536 > /dev/null 2>&1
has pmctag && tools/pmc.sh ${TAG}_tag_B16384 --scenario simple_tag --batch 16384 > /dev/null 2>&1
has pmcn64 && tools/pmc.sh ${TAG}_spread64_B4096 --agents 64 --batch 4096 > /dev/null 2>&1
ls $O
