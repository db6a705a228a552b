#!/bin/bash
# usage: tools/isa.sh <file-stem>   -> /tmp/isa/<stem>.s, prints per-kernel resource usage
cd /root/repo
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=14 -S --cuda-device-only -o /tmp/isa/$1.s multiagent_particle_envs_amd/csrc/mpe_$1.hip 2>&1 | grep -v "warning\|^$" | head
python3 - "$1" <<'PY'
import re,sys
s=open('/tmp/isa/%s.s'%sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", s, re.S):
    name=m.group(1)
    try:
        i=s.index("\n"+name+":"); j=s.index(".Lfunc_end", i); body=s[i:j]
        v=len(re.findall(r"^\s+v_",body,re.M)); sc=len(re.findall(r"^\s+s_",body,re.M)); ds=len(re.findall(r"^\s+ds_",body,re.M)); g=len(re.findall(r"^\s+(global|flat|buffer)_",body,re.M))
    except ValueError:
        v=sc=ds=g=-1
    print("%-90s sgpr %3s vgpr %3s | static valu %5d salu %5d ds %4d vmem %4d"%(name[:90],m.group(2),m.group(3),v,sc,ds,g))
PY
