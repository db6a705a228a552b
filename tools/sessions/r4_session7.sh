#!/bin/bash
# round 4, GPU visit 7: mpe_step_rows (World.step inside the row-program launch), mpe_episode_finish, full suite, rates
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s7}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 600 python -m pytest tests/test_rowspec.py -m gpu -x -q > $O/pytest_rowspec.log 2>&1; echo "rowspec rc=$?"; tail -25 $O/pytest_rowspec.log | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -15 $O/pytest_gpu.log | cut -c1-250
SC="corral,simple_spread,simple_adversary:num_agents=4:num_adversaries=2,simple_adversary:num_agents=6:num_adversaries=2,simple_world_comm:num_good_agents=2:num_adversaries=3,simple_world_comm:num_good_agents=3:num_adversaries=5,simple_adversary:num_agents=10:num_adversaries=3,simple_world_comm:num_good_agents=5:num_adversaries=6,simple_adversary:num_agents=30:num_adversaries=9"
timeout 600 python tools/rowspec_rate.py --scenarios "$SC" --eager-only --no-generic > $O/rate.txt 2> $O/rate.err; echo "rate rc=$?"; grep -v "^\[" $O/rate.txt; tail -3 $O/rate.err
