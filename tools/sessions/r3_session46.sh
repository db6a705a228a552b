#!/bin/bash
# last look: the in-tree library as it stands (rebuilt from HEAD), smoke + the ABI / parity files of the suite
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s46}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_abi.py tests/test_gpu_parity.py tests/test_gpu_rollout.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
