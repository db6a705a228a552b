cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 7 --warmup 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
python bench.py --gpus 1 --steps 33 --warmup 3 --no-cpu-baseline --mode eager 2>&1 | tail -1 | cut -c1-200
python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --mode fused 2>&1 | tail -1 | cut -c1-200
python bench.py --gpus 1 --steps 20 --warmup 2 --cpu-seconds 1 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(sorted(d.keys())); print(d['cpu_baseline'])"
