#!/bin/bash
# round 4, GPU visit 20: the full set -- GPU suite (+ parity_r4.json), smoke, the driver-style line, N=2 rehearsal on one GPU,
# kernel traces of every BASELINE config + the row-program step, PMC passes (C2 / C3 full; headline, 1M, C4 traffic; k_rows mix)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s20}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_r4.json
( time timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
cp $R/gpurun_out/parity_r4.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench rc=$?"; grep real $O/bench_20.time
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.3f G ms/step %.5f k_us %.3f frac %.3f frac_timed_region %.3f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_us_per_launch"], r["frac"], r["frac_timed_region"]))
    for kk,vv in d["extra"]["configs"].items(): print("  ", kk, "%.4g" % vv["value"], "k_us %.3f frac %.3f roll %.2f" % (vv["roofline"]["kernel_us_per_launch"], vv["roofline"]["frac"], vv["fused_rollout"]["kernel_us_per_step"]))
    v=d["extra"]["hbm_resident"]; print("  1M k_us %.2f frac %.3f" % (v["roofline"]["kernel_us_per_launch"], v["roofline"]["frac"]))
    print("  python_api %.3g  ids kernel %.2f" % (d["extra"]["python_api"]["value"], d["extra"]["int_action_ids"]["kernel_us_per_launch"]))
    print("  user_scenario", json.dumps(d["extra"].get("user_scenario"))[:600])
    print("  cpu_baseline", json.dumps(d["cpu_baseline"])[:300])
except Exception as e: print("parse failed", e)
PY
timeout 300 python bench.py --gpus 2 --all-ranks-on-gpu0 --steps 20 --warmup 5 --no-extra 2>$O/bench_n2.err | tail -1 > $O/bench_n2.json; python -c "
import json; d=json.load(open('$O/bench_n2.json')); print('N=2 rehearsal: n_gpus', d['n_gpus'], 'value %.3g' % d['value'], 'per_gpu', d.get('per_gpu_value'), d['config'].get('barrier_backend'), [r.get('device') for r in d['config'].get('ranks', [])][:2])"
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
# the row-program step (examples/corral.py and simple_spread's program) under the kernel trace
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_rows -o x -- \
    python $R/tools/rowspec_rate.py --scenarios corral,simple_spread --eager-only --no-generic --steps 200 > $O/trace_rows.log 2> $O/trace_rows.err)
python tools/trace_summary.py $(find $O/trace_rows -name "x_kernel_trace.csv" | head -1) "rocprofv3 --kernel-trace --stats -- python tools/rowspec_rate.py --scenarios corral,simple_spread --eager-only --no-generic --steps 200" > $O/rows_kernel_trace_summary.txt
cp $(find $O/trace_rows -name "x_kernel_stats.csv" | head -1) $O/rows_kernel_stats.csv 2>/dev/null; rm -rf $O/trace_rows
grep -v "^\[" $O/trace_rows.log | head; grep "k_rows" $O/rows_kernel_trace_summary.txt | head -6 | cut -c1-220
# counters
for cfg in "spread3_B4096 k_split --batch 4096" "tag_B16384 k_split --scenario simple_tag --batch 16384"; do
  set -- $cfg; name=$1; pat=$2; shift 2
  timeout 500 tools/pmc.sh ${TAG}_$name "$@" > /dev/null 2>&1
  python profiles/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$name $pat > $O/pmc_$name.txt 2>>$O/err.log
  rm -rf $R/gpurun_out/pmc_${TAG}_$name
  grep "traffic_bytes\|Kernel_Name" $O/pmc_$name.txt | cut -c1-200
done
export PMC_TRAFFIC_ONLY=1
for cfg in "spread3_B65536 k_split" "spread3_B1M k_split --batch 1048576 --steps 10" "spread64_B4096 k_duo --agents 64 --batch 4096 --steps 10"; do
  set -- $cfg; name=$1; pat=$2; shift 2
  timeout 400 tools/pmc.sh ${TAG}_$name "$@" > /dev/null 2>&1
  python profiles/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$name $pat > $O/pmc_$name.txt 2>>$O/err.log
  rm -rf $R/gpurun_out/pmc_${TAG}_$name
  grep "traffic_bytes\|Kernel_Name" $O/pmc_$name.txt | cut -c1-200
done
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_BRANCH"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/$name -o x -- python $R/tools/rowspec_rate.py --scenarios simple_spread --eager-only --no-generic --steps 60 > $O/pmc_rows_$name.log 2>&1
done
cd $R
python profiles/pmc_summary.py $O/pmc 'k_rows<true, true>' > $O/pmc_rows_spread3_B65536.txt 2>> $O/err.log; grep "traffic_bytes\|Kernel_Name\|SQ_INSTS_SALU\|SQ_INSTS_VALU\|SQ_WAVE_CYCLES\|SQ_WAIT_ANY" $O/pmc_rows_spread3_B65536.txt | cut -c1-200
rm -rf $O/pmc; rm -f $O/pmc_rows_*.log; tail -3 $O/err.log 2>/dev/null
for k in 1 2 3; do
  timeout 200 python bench.py --agents 64 --batch 4096 --steps 50 --warmup 10 --no-extra --no-cpu-baseline --region-ms 300 >> $O/c4_processes.jsonl 2>> $O/c4.err
done
python - <<PY
import json
for l in open("$O/c4_processes.jsonl"):
    d=json.loads(l); print("C4 process: k_us %.2f value %.4g frac %.3f | %s" % (d["roofline"]["kernel_us_per_launch"], d["value"], d["roofline"]["frac"], d["extra"]["box"].get("uuid")))
PY
ls $O
