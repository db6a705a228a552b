#!/bin/bash
# the driver's round-end sequence, rehearsed: GPU suite, smoke, the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s39}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
rm -f $R/gpurun_out/parity_r3.json
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cp $R/gpurun_out/parity_r3.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench rc=$?"; grep real $O/bench_20.time
python - <<PY
import json
d=json.loads(open("$O/bench_20.json").read().strip().splitlines()[-1])
print("value %.3f G k_us %.3f frac %.3f n_gpus %d" % (d["value"]/1e9, d["roofline"]["kernel_us_per_launch"], d["roofline"]["frac"], d["n_gpus"]))
for kk,vv in d["extra"]["configs"].items(): print("  ", kk, "k_us %.2f frac %.3f roll %.2f" % (vv["roofline"]["kernel_us_per_launch"], vv["roofline"]["frac"], vv["fused_rollout"]["kernel_us_per_step"]))
h=d["extra"]["hbm_resident"]; print("   1M k_us %.2f frac %.3f" % (h["roofline"]["kernel_us_per_launch"], h["roofline"]["frac"]))
print("   python_api %.3g  ids kernel %.2f" % (d["extra"]["python_api"]["value"], d["extra"]["int_action_ids"]["kernel_us_per_launch"]))
PY
timeout 300 python bench.py --gpus 2 --all-ranks-on-gpu0 --steps 20 --warmup 5 --no-extra 2>$O/bench_n2.err | tail -1 > $O/bench_n2.json; python -c "
import json; d=json.load(open('$O/bench_n2.json')); print('N=2 rehearsal: n_gpus', d['n_gpus'], 'value %.3g' % d['value'], d['config'].get('barrier_backend'), [r.get('device') for r in d['config'].get('ranks', [])][:2])"
