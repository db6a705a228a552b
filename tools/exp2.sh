cd /tmp
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rows -o x -- python $R/tools/probe_wide.py 64 4096 > /dev/null 2>&1
head -8 $(find $R/gpurun_out/prof_rows -name "x_kernel_stats.csv" | head -1) | cut -c1-160
