#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r3s35}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 300 python tools/ids_debug.py 2>&1 | grep -v amdgpu.ids | tee $O/ids.log
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["roofline"]["kernel_us_per_launch"], d["roofline"]["kernel_timing"])
print({k:v for k,v in d["extra"]["int_action_ids"].items() if k not in ("what","repeats")})
for k,v in d["extra"]["configs"].items(): print(k, v["roofline"]["kernel_us_per_launch"], v["roofline"].get("kernel_timing"))
h=d["extra"]["hbm_resident"]["roofline"]; print("1M", h["kernel_us_per_launch"], h.get("kernel_timing"))
PY
