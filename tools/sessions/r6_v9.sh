cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out/r6v9; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$PWD; R=$PWD
tools/sessions/_gpu_ok.sh || exit 0
timeout 120 python tools/server_profile.py 65536 1000 4 2>&1 | tail -5
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_served -o x -- python $R/tools/server_profile.py 65536 1000 8 > $O/trace_served.log 2>&1)
f=$(find $O/trace_served -name "x_kernel_trace.csv" | head -1)
python tools/trace_summary.py $f "rocprofv3 --kernel-trace --stats -- python tools/server_profile.py 65536 1000 8   (8 server launches of 1000 commanded steps each)" > $O/served_spread3_B65536_kernel_trace_summary.txt
cp $(find $O/trace_served -name "x_kernel_stats.csv" | head -1) $O/served_spread3_B65536_kernel_stats.csv; rm -rf $O/trace_served
head -8 $O/served_spread3_B65536_kernel_trace_summary.txt | cut -c1-220; tail -4 $O/trace_served.log
mkdir -p $O/pmc
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/$grp -o x -- python $R/tools/server_profile.py 65536 1000 4 > $O/pmc/$grp.log 2>&1); tail -2 $O/pmc/$grp.log
done
python profiles/pmc_summary.py $O/pmc "Lb1ELi2ELb0ELb1E" > $O/pmc_served_spread3_B65536.txt 2>&1 || python profiles/pmc_summary.py $O/pmc "k_split" > $O/pmc_served_spread3_B65536.txt
cat $O/pmc_served_spread3_B65536.txt | cut -c1-200
find $O/pmc -name "*.csv" | head; 
