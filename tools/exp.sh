cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do
for v in single two; do
  if [ $v = two ]; then unset MPE_WIDE_SINGLE_LAUNCH; else export MPE_WIDE_SINGLE_LAUNCH=1; fi
  echo "== $v"; python tools/probe_wide.py 64 4096 2>&1 | grep "full step\|observe only"
done; done
unset MPE_WIDE_SINGLE_LAUNCH
python tools/probe_wide.py 64 16384 2>&1 | grep "full step"
MPE_WIDE_SINGLE_LAUNCH=1 python tools/probe_wide.py 64 16384 2>&1 | grep "full step"
python tools/probe_wide.py 16 16384 2>&1 | grep "full step"
MPE_WIDE_SINGLE_LAUNCH=1 python tools/probe_wide.py 16 16384 2>&1 | grep "full step"
