cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
for v in base new; do
  if [ $v = new ]; then unset MPE_HIP_LIB; else export MPE_HIP_LIB=$PWD/tools/ubench/ablate/libmpe_base.so; fi
  echo "== $v"; python tools/probe_wide.py 64 4096 2>&1 | grep "full step\|observe only"
done; done
