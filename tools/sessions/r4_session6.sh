#!/bin/bash
# round 4, GPU visit 6: full GPU suite on the reduced kernel table (team sizes via row programs); team-size A/B (k_split entries vs programs); rates
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s6}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -40 $O/pytest_gpu.log | cut -c1-250
SC="simple_adversary:num_agents=4:num_adversaries=2,simple_adversary:num_agents=6:num_adversaries=2,simple_world_comm:num_good_agents=1:num_adversaries=2,simple_world_comm:num_good_agents=2:num_adversaries=3,simple_world_comm:num_good_agents=3:num_adversaries=5"
MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_teamgrid.so timeout 600 python tools/rowspec_rate.py --scenarios "$SC" --eager-only --no-generic > $O/team_ab.txt 2> $O/team_ab.err; echo "team A/B rc=$?"; grep -v "^\[" $O/team_ab.txt; tail -3 $O/team_ab.err
timeout 600 python tools/rowspec_rate.py --scenarios "simple_adversary:num_agents=10:num_adversaries=3,simple_world_comm:num_good_agents=5:num_adversaries=6,simple_adversary:num_agents=30:num_adversaries=9" --eager-only > $O/team_big.txt 2> $O/team_big.err; echo "big teams rc=$?"; grep -v "^\[" $O/team_big.txt; tail -3 $O/team_big.err
