cd $GRAFT_REPO_ROOT
python -m pytest tests/test_f3_scenarios.py -m gpu -x -q 2>&1 | tail -3
for sc in simple_push simple_adversary; do
python bench.py --scenario $sc --batch 65536 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['config']['workload'][:40], '%.3g steps/s'%d['value'], '%.2f us/step'%(d['ms_per_step']*1e3), 'frac %.3f'%r['frac'], r['algorithmic_bytes_per_env_step'], 'fused', d['extra']['fused_rollout']['kernel_us_per_step'])"
done
