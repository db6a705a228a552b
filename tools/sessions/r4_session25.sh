#!/bin/bash
# round 4, GPU visit 25: done programs + episodes that end inside the step launch (mpe_step_rows_episode); compiled programs with the
# op loops really unrolled (-pragma-unroll-threshold); costs and rates again
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s25}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 900 python -m pytest tests/test_rowspec.py -m gpu -x -q > $O/pytest_rowspec.log 2>&1; echo "rowspec rc=$?"; tail -12 $O/pytest_rowspec.log | cut -c1-300
timeout 600 python tools/finish_cost.py > $O/finish_cost.txt 2> $O/finish_cost.err; echo "finish_cost rc=$?"; cat $O/finish_cost.txt; grep -v amdgpu.ids $O/finish_cost.err | tail -5
timeout 600 python tools/rows_ablate.py > $O/rows_ablation.txt 2> $O/rows_ablation.err; echo "ablate rc=$?"; cat $O/rows_ablation.txt; grep -v amdgpu.ids $O/rows_ablation.err | tail -5
SC="corral,simple_spread,simple_tag,simple_adversary:num_agents=4:num_adversaries=2,simple_adversary:num_agents=6:num_adversaries=2,simple_world_comm:num_good_agents=2:num_adversaries=3,simple_world_comm:num_good_agents=3:num_adversaries=5,simple_adversary:num_agents=10:num_adversaries=3,simple_world_comm:num_good_agents=5:num_adversaries=6"
timeout 900 python tools/rowspec_rate.py --scenarios "$SC" --eager-only --no-generic --compiled > $O/rate.txt 2> $O/rate.err; echo "rate rc=$?"; grep -v "^\[" $O/rate.txt; grep -v amdgpu.ids $O/rate.err | tail -12
