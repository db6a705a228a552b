cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for ch in 1 2; do
  MPE_SPLIT_CHUNKS=$ch python bench.py --no-cpu-baseline --no-extra --repeats 3 > gpurun_out/exp_b.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/exp_b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('chunks $ch: us/launch %.2f'%r['kernel_us_per_launch'])"
done; done
for ch in 1 2; do
  MPE_SPLIT_CHUNKS=$ch python bench.py --no-cpu-baseline --no-extra --repeats 3 --batch 1048576 --steps 200 > gpurun_out/exp_b.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/exp_b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('1M chunks $ch: us/launch %.2f'%r['kernel_us_per_launch'])"
done
