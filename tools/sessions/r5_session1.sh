#!/bin/bash
# round 5, GPU visit 1: baseline of the round-4 HEAD (GPU suite) + the headline cut into S independent sub-batches on S HIP streams
# (bench.py --streams S; StreamedRollout: the dependent-launch gap of one chain under the kernel of the other)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s1}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
for S in 1 2 4 1 2; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --streams $S > $O/bench_S$S.json 2> $O/bench_S$S.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_S$S.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("streams $S: value %.3f G  us/step %.3f  k_us %.3f frac %.3f frac_timed %.3f" % (d["value"]/1e9, d["ms_per_step"]*1e3, r["kernel_us_per_launch"], r["frac"], r["frac_timed_region"]))
except Exception as e: print("streams $S parse failed", e)
PY
done
for B in 4096 16384; do for S in 1 2 4; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --batch $B --streams $S > $O/bench_B${B}_S$S.json 2> $O/bench_B${B}_S$S.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_B${B}_S$S.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("B $B streams $S: value %.3f G  us/step %.3f  k_us %.3f" % (d["value"]/1e9, d["ms_per_step"]*1e3, r["kernel_us_per_launch"]))
except Exception as e: print("B $B streams $S parse failed", e)
PY
done; done
( time timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300; grep real $O/pytest.time
