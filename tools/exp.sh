cd $GRAFT_REPO_ROOT
python tools/probe_dvfs.py 64 4096 3 2>&1 | grep -v amdgpu.ids
sleep 2
python tools/probe_dvfs.py 3 65536 2 2>&1 | grep -v amdgpu.ids
