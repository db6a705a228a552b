#!/bin/bash
# round 5, GPU visit 22: the full-size traced test with the knife-edge explanation
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r5s22}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
timeout 600 python -m pytest tests/test_gpu_traced.py -m gpu -q 2>&1 | tail -4
exit 0
