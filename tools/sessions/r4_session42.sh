#!/bin/bash
# round 4, GPU visit 42: final code (five-entry images, episode rollouts) -- GPU suite (+ parity_r4.json), smoke, the driver-style line, N=2 rehearsal, kernel trace and PMC of
# the row-program step (interpreted and compiled in), finish cost
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s42}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_r4.json
( time timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
cp $R/gpurun_out/parity_r4.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
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
    for k in ("program","compiled","generic","compiled_fused_rollout"): print("  user_scenario", k, "%.4g" % u[k]["value"], "%.2f us" % u[k]["us_per_step"], u[k]["path"])
    for k in ("program_graph","compiled_graph"): print("  user_scenario", k, "%.4g" % u[k]["value"], "%.3f us" % (u[k]["ms_per_step"]*1e3))
except Exception as e: print("parse failed", e)
PY
timeout 300 python bench.py --gpus 2 --all-ranks-on-gpu0 --steps 20 --warmup 5 --no-extra 2>$O/bench_n2.err | tail -1 > $O/bench_n2.json; python -c "
import json; d=json.load(open('$O/bench_n2.json')); print('N=2 rehearsal: n_gpus', d['n_gpus'], 'value %.3g' % d['value'], d['config'].get('barrier_backend'))"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_rows -o x -- \
    python $R/tools/rowspec_rate.py --scenarios corral,simple_spread --eager-only --no-generic --compiled --steps 200 > $O/trace_rows.log 2> $O/trace_rows.err)
python tools/trace_summary.py $(find $O/trace_rows -name "x_kernel_trace.csv" | head -1) "rocprofv3 --kernel-trace --stats -- python tools/rowspec_rate.py --scenarios corral,simple_spread --eager-only --no-generic --compiled --steps 200" > $O/rows_kernel_trace_summary.txt
cp $(find $O/trace_rows -name "x_kernel_stats.csv" | head -1) $O/rows_kernel_stats.csv 2>/dev/null; rm -rf $O/trace_rows
grep "k_rows\|mpe_rows_" $O/rows_kernel_trace_summary.txt | head -8 | cut -c1-220
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_BRANCH"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/$name -o x -- python $R/tools/rowspec_rate.py --scenarios simple_spread --eager-only --no-generic --compiled --steps 60 > $O/pmc_rows_$name.log 2>&1
done
cd $R
python profiles/pmc_summary.py $O/pmc 'k_rows<true, false>' > $O/pmc_rows_spread3_B65536.txt 2>> $O/err.log
python profiles/pmc_summary.py $O/pmc 'end:_s' > $O/pmc_rows_compiled_spread3_B65536.txt 2>> $O/err.log
for f in $O/pmc_rows_spread3_B65536.txt $O/pmc_rows_compiled_spread3_B65536.txt; do grep "traffic_bytes\|Kernel_Name\|SQ_INSTS_SALU\|SQ_INSTS_VALU\|SQ_WAVE_CYCLES\|SQ_WAIT_ANY\|VGPR" $f | cut -c1-200; done
rm -rf $O/pmc; rm -f $O/pmc_rows_*.log; tail -3 $O/err.log 2>/dev/null
timeout 600 python tools/finish_cost.py > $O/finish_cost.txt 2> $O/finish_cost.err; echo "finish_cost rc=$?"; grep "env.step\|from Python" $O/finish_cost.txt | cut -c1-220
ls $O
timeout 600 python tools/rows_rollout_rate.py > $O/rollout_rate.txt 2> $O/rollout_rate.err; echo "rollout rate rc=$?"; cat $O/rollout_rate.txt
