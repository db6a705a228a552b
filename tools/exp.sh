cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -12
python tools/probe_wide.py 16 16384 2>&1 | grep "full step\|obs fill"
python tools/probe_wide.py 8 65536 2>&1 | grep "full step\|obs fill"
python tools/probe_wide.py 32 8192 2>&1 | grep "full step\|obs fill"
