cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for n in 0 0; do
  python bench.py --no-cpu-baseline --repeats 3 > gpurun_out/exp_b.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/exp_b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('us/launch %.2f'%r['kernel_us_per_launch'], 'fused', d['extra']['fused_rollout']['kernel_us_per_step'])"
done
python bench.py --no-cpu-baseline --scenario simple_tag --batch 16384 --repeats 3 > gpurun_out/exp_t.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/exp_t.json').read().strip().splitlines()[-1]); r=d['roofline']; print('tag us/launch %.2f'%r['kernel_us_per_launch'], 'fused', d['extra']['fused_rollout']['kernel_us_per_step'])"
