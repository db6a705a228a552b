#!/bin/bash
# round 4, GPU visit 3: row programs (mpe_rows) -- bit-identity with the fused kernels, the custom scenario, rates; device stamps v3 (one deferred store per wave)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s3}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_rowspec.py -m gpu -x -q > $O/pytest_rowspec.log 2>&1; echo "rowspec rc=$?"; tail -25 $O/pytest_rowspec.log
timeout 600 python tools/rowspec_rate.py > $O/rowspec_rate.txt 2> $O/rowspec_rate.err; echo "rate rc=$?"; grep -v "^\[" $O/rowspec_rate.txt | head -40; tail -5 $O/rowspec_rate.err
timeout 900 python tools/device_span.py C2 C3 C5 --out $O > $O/span.log 2> $O/span.err; echo "span rc=$?"; grep -E "^# device|period|span |gap|slope|device period" $O/span.log; tail -5 $O/span.err
