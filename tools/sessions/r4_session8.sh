#!/bin/bash
# round 4, GPU visit 8: k_rows without a reward wave (per-agent reward programs, op prefetch); episode finish; rates; a kernel trace of the program step
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s8}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 600 python -m pytest tests/test_rowspec.py tests/test_f3_scenarios.py -m gpu -q -k "rowspec or team or shape or program or specs or done_callback" > $O/pytest_rowspec.log 2>&1; echo "rowspec+teams rc=$?"; tail -12 $O/pytest_rowspec.log | cut -c1-300
SC="corral,simple_spread,simple_tag,simple_adversary:num_agents=4:num_adversaries=2,simple_adversary:num_agents=6:num_adversaries=2,simple_world_comm:num_good_agents=2:num_adversaries=3,simple_world_comm:num_good_agents=3:num_adversaries=5,simple_adversary:num_agents=10:num_adversaries=3,simple_world_comm:num_good_agents=5:num_adversaries=6,simple_adversary:num_agents=30:num_adversaries=9"
timeout 600 python tools/rowspec_rate.py --scenarios "$SC" --eager-only --no-generic > $O/rate.txt 2> $O/rate.err; echo "rate rc=$?"; grep -v "^\[" $O/rate.txt; tail -3 $O/rate.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o rows -- python $R/tools/rowspec_rate.py --scenarios corral,simple_spread --steps 200 --eager-only --no-generic > $O/prof.log 2>&1; echo "prof rc=$?"
cd $R; python - <<PY
import csv, glob
for f in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:8]:
        print("%-100s calls %6s avg %8.2f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
