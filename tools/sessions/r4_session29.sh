#!/bin/bash
# round 4, GPU visit 29: (moves, utterances) tuple actions for the communication scenarios; smoke() with the row-program check; rates
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s29}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python -m pytest tests/test_rowspec.py -m gpu -x -q -k "pair_of_tensors or episodes or horizon" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.log | cut -c1-300
SC="simple_speaker_listener,simple_reference,simple_crypto,simple_world_comm,simple_world_comm:num_good_agents=2:num_adversaries=3,simple_world_comm:num_good_agents=3:num_adversaries=5,simple_world_comm:num_good_agents=5:num_adversaries=6"
timeout 900 python tools/rowspec_rate.py --scenarios "$SC" --eager-only --no-generic --compiled > $O/rate.txt 2> $O/rate.err; echo "rate rc=$?"; grep -v "^\[" $O/rate.txt; grep -v amdgpu.ids $O/rate.err | tail -3
