#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s6}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_race.py tests/test_f3_scenarios.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
tools/ab_matrix.sh $TAG "tag:3:16384 spread:3:4096 spread:3:65536 simple_adversary:3:65536 simple:1:65536 spread:4:16384 simple_push:2:65536 tag:3:65536" nodual base
export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_clk.so
timeout 120 python tools/phase_clock.py simple_tag 16384 roll 2>>$O/err.log | tee -a $O/phase.txt
timeout 120 python tools/phase_clock.py simple_spread 4096 roll 2>>$O/err.log | tee -a $O/phase.txt
unset MPE_HIP_LIB
tail -3 $O/err.log
