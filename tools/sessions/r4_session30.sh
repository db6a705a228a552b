#!/bin/bash
# round 4, GPU visit 30: step()'s short path for (moves, utterances); smoke(); rates of the communication scenarios
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s30}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python -m pytest tests/test_rowspec.py tests/test_f3_scenarios.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.log | cut -c1-300
SC="simple_speaker_listener,simple_reference,simple_crypto,simple_world_comm,simple_world_comm:num_good_agents=3:num_adversaries=5"
timeout 900 python tools/rowspec_rate.py --scenarios "$SC" --eager-only --no-generic --compiled > $O/rate.txt 2> $O/rate.err; echo "rate rc=$?"; grep -v "^\[" $O/rate.txt; grep -v amdgpu.ids $O/rate.err | tail -3
