cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/probe_wide.py 64 4096 2>&1 | grep -v amdgpu.ids
