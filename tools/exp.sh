cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1g
python bench.py > gpurun_out/r1g/bench_default.json 2> gpurun_out/r1g/bench_default.err
python bench.py --no-cpu-baseline --agents 64 --batch 4096 --steps 200 > gpurun_out/r1g/bench_n64.json 2>/dev/null
python bench.py --no-cpu-baseline --scenario simple_tag --batch 16384 > gpurun_out/r1g/bench_tag.json 2>/dev/null
python bench.py --no-cpu-baseline --batch 1048576 --steps 200 > gpurun_out/r1g/bench_1M.json 2>/dev/null
python bench.py --no-cpu-baseline --mode api --no-extra > gpurun_out/r1g/bench_api.json 2>/dev/null
python __graft_entry__.py smoke 2>&1 | tail -4
