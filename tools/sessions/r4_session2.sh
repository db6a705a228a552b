#!/bin/bash
# round 4, GPU visit 2: device-clock stamps with both end-stamp flavours; reference-style scenario files on the HIP physics; full GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s2}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python tools/device_span.py C2 C3 C5 1M --out $O > $O/span.log 2> $O/span.err; echo "span rc=$?"; grep -E "^#|period|span |gap|slope|cross|device period" $O/span.log; tail -5 $O/span.err
timeout 600 python -m pytest tests/test_refstyle.py -m gpu -x -q > $O/pytest_refstyle.log 2>&1; echo "refstyle rc=$?"; tail -15 $O/pytest_refstyle.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -8 $O/pytest_gpu.log
