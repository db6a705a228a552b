#!/bin/bash
# round 4, GPU visit 9: parity at size (fp32 comparator, C4 sample, reference N=64 worlds in the real grid), full GPU suite, VMM experiment for C4
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r4s9}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "drift or spread64 or reference_worlds" > $O/pytest_parity.log 2>&1; echo "parity rc=$?"; grep -E "free-running|passed|failed|Error|assert" $O/pytest_parity.log | cut -c1-400 | tail -20
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -8 $O/pytest_gpu.log | cut -c1-250
cp gpurun_out/parity_r4.json $O/parity.json 2>/dev/null
timeout 600 python tools/vmm_c4.py 36 > $O/vmm_c4.txt 2> $O/vmm_c4.err; echo "vmm rc=$?"; cat $O/vmm_c4.txt | cut -c1-200; tail -5 $O/vmm_c4.err
