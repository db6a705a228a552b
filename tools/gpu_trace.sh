#!/bin/bash
# rocprofv3 kernel trace of a short default bench run -> gpurun_out/<tag>/ (stats csv + per-dispatch trace)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-trace}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o x -- python $R/bench.py --no-cpu-baseline --no-extra --repeats 2 "$@" > $O/bench.json 2> $O/bench.err
find $O/prof -name "*.csv" | head; 
python - <<PY
import csv,glob,collections
f=glob.glob("$O/prof/**/x_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# durations and gaps for the dominant kernel in the last 500 dispatches
dur=collections.defaultdict(list); gaps=[]
prev=None
for r in rows[-2000:]:
    n=r["Kernel_Name"][:60]; s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    dur[n].append(e-s)
    if prev is not None: gaps.append(s-prev)
    prev=e
for n,v in dur.items():
    v.sort(); print("%-62s n=%5d med=%7.2f us p10=%7.2f p90=%7.2f"%(n,len(v),v[len(v)//2]/1e3,v[len(v)//10]/1e3,v[9*len(v)//10]/1e3))
gaps.sort(); print("gap between consecutive dispatches: med=%.2f us p10=%.2f p90=%.2f"%(gaps[len(gaps)//2]/1e3,gaps[len(gaps)//10]/1e3,gaps[9*len(gaps)//10]/1e3))
PY
