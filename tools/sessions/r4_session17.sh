#!/bin/bash
# round 4, GPU visit 17: PMC counters of k_rows (instruction mix, waits, LDS conflicts) on the spread program step
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s17}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_BRANCH"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc/$name -o x -- python $R/tools/rowspec_rate.py --scenarios simple_spread --eager-only --no-generic --steps 60 > $O/pmc_$name.log 2>&1
done
cd $R
python profiles/pmc_summary.py $O/pmc 'k_rows<true, true>' > $O/pmc_rows_spread3_B65536.txt 2>> $O/err.log; cat $O/pmc_rows_spread3_B65536.txt
python profiles/pmc_summary.py $O/pmc k_split > $O/pmc_split_spread3_B65536.txt 2>> $O/err.log; cat $O/pmc_split_spread3_B65536.txt
rm -rf $O/pmc; tail -3 $O/err.log
