#!/bin/bash
# round 4, GPU visit 4: row programs -- bit-identity tests again; kernel-trace split of the two launches
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s4}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_rowspec.py -m gpu -q > $O/pytest_rowspec.log 2>&1; echo "rowspec rc=$?"; tail -40 $O/pytest_rowspec.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o rows -- python $R/tools/rowspec_rate.py --scenarios corral --steps 200 > $O/prof.log 2>&1; echo "prof rc=$?"
cd $R; python - <<PY
import csv, glob
for f in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:12]:
        print("%-90s calls %6s avg %8.2f us  total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
