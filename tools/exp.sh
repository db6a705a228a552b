cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --no-cpu-baseline --agents 64 --batch 4096 --steps 100 --repeats 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('n64 per-step %.2f us, frac %.3f; fused'%(r['kernel_us_per_launch'], r['frac']), d.get('extra',{}).get('fused_rollout'))"
