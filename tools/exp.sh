cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/probe_host.py 2>&1 | grep "step()"
python bench.py --no-cpu-baseline --mode api --no-extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('api mode: %.3g steps/s, %.2f us/step'%(d['value'], d['ms_per_step']*1e3))"
