#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s42}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python bench.py --gpus 2 --all-ranks-on-gpu0 --steps 20 --warmup 5 --no-extra --backend auto 2>$O/n2.err | tail -1 > $O/n2.json
python -c "
import json; d=json.load(open('$O/n2.json')); print('N=2 auto:', d['n_gpus'], '%.3g' % d['value'], d['config'].get('barrier_backend'), d['config'].get('barrier_note'))"
grep -i "rccl\|nccl" $O/n2.err | head -5
