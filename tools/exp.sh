cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py > gpurun_out/exp_default.json 2>gpurun_out/exp_default.err; tail -c 1500 gpurun_out/exp_default.json
