#!/bin/bash
# round 4, GPU visit 33: the block draw of the benchmark's moves with 16-byte stores: tests, its duration, the headline line
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s33}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_abi.py tests/test_gpu_rollout.py -m gpu -x -q > $O/pytest.log 2>&1; echo "abi+rollout rc=$?"; tail -3 $O/pytest.log | cut -c1-300
for k in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_$k.json 2> $O/bench_$k.err
  python -c "
import json; d=json.loads(open('$O/bench_$k.json').read().strip().splitlines()[-1]); print('value %.4g ms/step %.5f k_us %.3f frac_timed_region %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_us_per_launch'], d['roofline']['frac_timed_region']))"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o x -- python $R/bench.py --no-cpu-baseline --no-extra --repeats 2 --region-ms 40 --steps 200 > $O/trace.bench.json 2> $O/trace.err)
python tools/trace_summary.py $(find $O/trace -name "x_kernel_trace.csv" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra --repeats 2 --region-ms 40 --steps 200" > $O/spread3_B65536_kernel_trace_summary.txt
cp $(find $O/trace -name "x_kernel_stats.csv" | head -1) $O/spread3_B65536_kernel_stats.csv 2>/dev/null; rm -rf $O/trace
grep "k_random_actions\|k_split\|k_reset" $O/spread3_B65536_kernel_trace_summary.txt | cut -c1-200
