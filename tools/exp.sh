cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --no-cpu-baseline > gpurun_out/exp_b.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/exp_b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('spread3 %.3g'%d['value'], 'us/launch %.2f'%r['kernel_us_per_launch'], 'frac %.3f'%r['frac'], 'fused us/step', d['extra']['fused_rollout']['kernel_us_per_step'])"
python bench.py --no-cpu-baseline --scenario simple_tag --batch 16384 > gpurun_out/exp_t.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/exp_t.json').read().strip().splitlines()[-1]); r=d['roofline']; print('tag %.3g'%d['value'], 'us/launch %.2f'%r['kernel_us_per_launch'], 'frac %.3f'%r['frac'], 'fused us/step', d['extra']['fused_rollout']['kernel_us_per_step'])"
