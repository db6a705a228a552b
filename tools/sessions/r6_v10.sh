cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out/r6v10; mkdir -p $O/pmc; export TMPDIR=/tmp PYTHONPATH=$PWD; R=$PWD
tools/sessions/_gpu_ok.sh || exit 0
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/$grp -o x -- python $R/tools/server_profile.py 65536 1000 4 > $O/pmc/$grp.log 2>&1); grep "launch " $O/pmc/$grp.log | tail -2
done
python profiles/pmc_summary.py $O/pmc "true, 2, false, true>" > $O/pmc_served_spread3_B65536.txt; cat $O/pmc_served_spread3_B65536.txt | cut -c1-200
rm -rf $O/pmc/*/x_kernel_trace.csv
