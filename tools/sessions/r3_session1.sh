#!/bin/bash
# round 3, GPU visit 1: the new bench.py paths (self-spawned ranks, RCCL fallback, 2 s regions, box fingerprint)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3s1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q > $O/pytest_multirank.log 2>&1; echo "multirank rc=$?"; tail -5 $O/pytest_multirank.log
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/bench_20.time; echo "bench20 rc=$?"; tail -3 $O/bench_20.time; tail -c 600 $O/bench_20.err
( time timeout 600 python bench.py --steps 1000 --warmup 50 --no-extra --no-cpu-baseline > $O/bench_1000.json 2> $O/bench_1000.err ) 2> $O/bench_1000.time; echo "bench1000 rc=$?"; tail -3 $O/bench_1000.time
python - <<PY
import json
for f in ("bench_20","bench_1000"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "value %.3f G ms/step %.5f timed_steps %s region_s %.3f k_us %.3f frac %.3f" % (d["value"]/1e9, d["ms_per_step"], d["config"]["timed_steps"], d["config"]["timed_region_s"], d["roofline"]["kernel_us_per_launch"], d["roofline"]["frac"]), d["repeats"])
        for k,v in d.get("extra",{}).items():
            if k=="configs":
                for kk,vv in v.items(): print("  ", kk, "%.4g" % vv["value"], "k_us %.2f frac %.3f" % (vv["roofline"]["kernel_us_per_launch"], vv["roofline"]["frac"]), "| roll k_us %.2f frac_c %.3f" % (vv["fused_rollout"]["kernel_us_per_step"], vv["fused_rollout"]["frac_compulsory"]), vv["repeats"])
            elif k=="hbm_resident": print("  1M: %.4g k_us %.2f frac %.3f" % (v["value"], v["roofline"]["kernel_us_per_launch"], v["roofline"]["frac"]), v["repeats"])
            elif k=="box": print("  box", json.dumps(v)[:1500])
            else: print("  ", k, "%.4g" % v["value"], v.get("kernel_us_per_step"))
        print("  cpu", d.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(f, "parse failed", e)
PY
rocm-smi --showclocks --showpower --showmaxpower --showmemorypartition --showcomputepartition --showdriverversion --showperflevel --json > $O/rocm_smi.json 2>&1
head -c 1500 $O/rocm_smi.json
