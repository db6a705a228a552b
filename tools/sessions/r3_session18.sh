#!/bin/bash
# team sizes beyond the reference's make_world: parity tests of the new k_split table entries, then their kernel times
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s18}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_f3_scenarios.py tests/test_gpu_rollout.py tests/test_gpu_abi.py -m gpu -x -q \
  -k "team_size or other_team or fused_rollout_equals_stepwise or abi" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
cp gpurun_out/parity_r3.json $O/parity_shapes.json 2>/dev/null
timeout 600 python tools/ab_kernels.py simple_adversary:3:65536 simple_adversary:2:65536:num_agents=2 \
  simple_adversary:4:65536:num_agents=4,num_adversaries=2 simple_adversary:6:65536:num_agents=6,num_adversaries=2 \
  simple_world_comm:6:65536 simple_world_comm:3:65536:num_good_agents=1,num_adversaries=2 \
  simple_world_comm:5:65536:num_good_agents=2,num_adversaries=3 simple_world_comm:8:65536:num_good_agents=3,num_adversaries=5 \
  > $O/shapes_perf.log 2>&1
cat $O/shapes_perf.log
