#!/bin/bash
# Same-box A/B of library variants: tools/ab_run.sh <tag> "<bench args>" <variant> [<variant> ...]   (variant "base" = the normal build)
# Each variant runs in `reps` (default 3) separate processes; prints kernel_us_per_launch and value per run.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; ARGS=$2; shift 2
REPS=${REPS:-3}
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    if [ $v == base ]; then unset MPE_HIP_LIB; else export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_$v.so; fi
    timeout 300 python bench.py --no-cpu-baseline --no-extra --repeats 3 $ARGS > $O/${v}_$rep.json 2> $O/${v}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/${v}_$rep.json").read().strip().splitlines()[-1])
    print("%-10s rep $rep  kernel_us %8.3f  frac %.3f  value %.4g  region_us/step %.3f" % ("$v", d["roofline"]["kernel_us_per_launch"], d["roofline"]["frac"], d["value"], d["roofline"]["timed_region_us_per_step"]))
except Exception as e:
    print("$v rep $rep failed", e, open("$O/${v}_$rep.err").read()[-400:])
PY
  done
done
