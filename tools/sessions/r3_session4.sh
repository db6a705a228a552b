#!/bin/bash
# round 3, GPU visit 4: full parity suite (pipelined rollouts in), A/B pipelined vs not, phase clocks, C4 diagnosis
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s4}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export TMPDIR=/tmp
rm -f $R/gpurun_out/parity_r3.json
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
cp $R/gpurun_out/parity_r3.json $O/ 2>/dev/null
tools/ab_matrix.sh $TAG "tag:3:16384 spread:3:4096 spread:3:65536 simple_adversary:3:65536 simple:1:65536 spread:4:16384" nopipe base
export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_clk.so
timeout 120 python tools/phase_clock.py simple_tag 16384 roll 2>>$O/err.log | tee -a $O/phase.txt
timeout 120 python tools/phase_clock.py simple_spread 4096 roll 2>>$O/err.log | tee -a $O/phase.txt
unset MPE_HIP_LIB
tools/c4_diag.sh ${TAG}_c4 3
tail -3 $O/err.log
