cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r2roll; mkdir -p $O; export TMPDIR=/tmp
tr() { name=$1; shift; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$name -o x -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --repeats 2 --mode fused "$@" > $O/$name.bench.json 2> $O/$name.err); cp $(find $O/t_$name -name "x_kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv; rm -rf $O/t_$name; head -3 $O/${name}_kernel_stats.csv; }
tr rollout_spread3_B65536 --steps 200 --warmup 25
tr rollout_tag_B16384 --scenario simple_tag --batch 16384 --steps 200 --warmup 25
tr rollout_spread64_B4096 --agents 64 --batch 4096 --steps 100 --warmup 25
