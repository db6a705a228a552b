cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1y
python bench.py > gpurun_out/r1y/bench_default.json 2> gpurun_out/r1y/bench_default.err
python bench.py --no-cpu-baseline --agents 64 --batch 4096 --steps 200 > gpurun_out/r1y/bench_n64.json 2>/dev/null
python bench.py --no-cpu-baseline --scenario simple_tag --batch 16384 > gpurun_out/r1y/bench_tag.json 2>/dev/null
python bench.py --no-cpu-baseline --batch 1048576 --steps 200 > gpurun_out/r1y/bench_1M.json 2>/dev/null
python bench.py --no-cpu-baseline --mode api --no-extra > gpurun_out/r1y/bench_api.json 2>/dev/null
for sc in simple_adversary simple_push simple_speaker_listener simple_reference simple_crypto simple_world_comm; do
  python bench.py --no-cpu-baseline --no-extra --scenario $sc > gpurun_out/r1y/bench_$sc.json 2>/dev/null
done
