cd $GRAFT_REPO_ROOT
for sc in simple_speaker_listener simple_reference simple_crypto simple_world_comm; do
python bench.py --scenario $sc --batch 65536 --repeats 3 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['config']['workload'][:44], '%.3g steps/s'%d['value'], '%.2f us/step'%(d['ms_per_step']*1e3), r.get('frac'), r.get('algorithmic_bytes_per_env_step'))"
done
