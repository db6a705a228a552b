#!/bin/bash
# round 4, GPU visit 19: range loops four entities per LDS round trip: tests, phase clocks, ablation, rates
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s19}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 600 python -m pytest tests/test_rowspec.py tests/test_f3_scenarios.py -m gpu -q -k "rowspec or team or shape or program or specs or done_callback" > $O/pytest_rowspec.log 2>&1; echo "rowspec+teams rc=$?"; tail -12 $O/pytest_rowspec.log | cut -c1-300
for s in simple_spread simple_tag; do
  MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_rowsclock.so timeout 200 python tools/rows_clock.py $s >> $O/rows_clock.txt 2>> $O/rows_clock.err
done
cat $O/rows_clock.txt | cut -c1-200; tail -3 $O/rows_clock.err
timeout 600 python tools/rows_ablate.py > $O/rows_ablation.txt 2> $O/rows_ablation.err; echo "ablate rc=$?"; cat $O/rows_ablation.txt; tail -5 $O/rows_ablation.err
SC="corral,simple_spread,simple_tag,simple_adversary:num_agents=4:num_adversaries=2,simple_adversary:num_agents=6:num_adversaries=2,simple_world_comm:num_good_agents=2:num_adversaries=3,simple_world_comm:num_good_agents=3:num_adversaries=5,simple_adversary:num_agents=10:num_adversaries=3,simple_world_comm:num_good_agents=5:num_adversaries=6,simple_adversary:num_agents=30:num_adversaries=9"
timeout 600 python tools/rowspec_rate.py --scenarios "$SC" --eager-only --no-generic > $O/rate.txt 2> $O/rate.err; echo "rate rc=$?"; grep -v "^\[" $O/rate.txt; tail -3 $O/rate.err
