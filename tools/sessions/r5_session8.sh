#!/bin/bash
# round 5, GPU visit 8: the traced kernels beside the hand-fused ones (the reference's own files, from their committed traces), a
# kernel trace + PMC passes of the traced convoy step, the bench line with the traced leg's roofline entry
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s8}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
timeout 900 python tools/refstyle_rate.py --json simple_spread --json simple_tag --json simple_adversary --json simple_push --json simple_reference --json simple_speaker_listener --json simple_crypto --json simple_world_comm --json simple tests/refstyle/convoy.py > $O/traced_vs_fused.txt 2> $O/traced_vs_fused.err; echo "traced_vs_fused rc=$?"; cat $O/traced_vs_fused.txt; tail -5 $O/traced_vs_fused.err | grep -v amdgpu
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o x -- \
    python $R/tools/refstyle_rate.py --profile-steps 300 --json simple_spread $R/tests/refstyle/convoy.py > $O/trace.log 2> $O/trace.err)
CMD="rocprofv3 --kernel-trace --stats -- python tools/refstyle_rate.py --profile-steps 300 --json simple_spread tests/refstyle/convoy.py"
python tools/trace_summary.py $(find $O/trace -name "x_kernel_trace.csv" | head -1) "$CMD" > $O/traced_kernel_trace_summary.txt
cp $(find $O/trace -name "x_kernel_stats.csv" | head -1) $O/traced_kernel_stats.csv 2>/dev/null; rm -rf $O/trace
grep "mpe_rows_\|k_rows\|Kernel\|#" $O/traced_kernel_trace_summary.txt | head -12 | cut -c1-240
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/$name -o x -- python $R/tools/refstyle_rate.py --profile-steps 60 $R/tests/refstyle/convoy.py > $O/pmc_$name.log 2>&1
done
cd $R
python profiles/pmc_summary.py $O/pmc 'end:_s' > $O/pmc_traced_convoy_B65536.txt 2>> $O/err.log
grep "traffic_bytes\|Kernel_Name\|SQ_INSTS_SALU\|SQ_INSTS_VALU\|SQ_WAVE_CYCLES\|SQ_WAIT_ANY\|VGPR\|LDS" $O/pmc_traced_convoy_B65536.txt | cut -c1-200
rm -rf $O/pmc; rm -f $O/pmc_*.log; tail -3 $O/err.log 2>/dev/null
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench rc=$?"; grep real $O/bench_20.time
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.3f G ms/step %.5f k_us %.3f frac %.3f frac_timed_region %.3f" % (d["value"]/1e9, d["ms_per_step"], r["kernel_us_per_launch"], r["frac"], r["frac_timed_region"]))
    u=d["extra"]["reference_style_file"]
    print("  roofline", {k: u["roofline"][k] for k in ("kernel_us_per_launch","algorithmic_bytes_per_env_step","achieved","frac")})
    for k in ("traced","traced_graph","traced_fused_rollout","host_path"): print("  reference_style_file", k, "%.4g" % u[k]["value"])
except Exception as e: print("parse failed", repr(e))
PY
exit 0
