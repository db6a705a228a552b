#!/bin/bash
# C4 (simple_spread N=64, B=4096) is 68 us in one process and 75 / 82 us in the next on the same box.
# N processes, each under `rocprofv3 --pmc` with one counter group: per process the k_duo duration (median over its dispatches)
# next to address-translation and fabric write-stall counters per dispatch -> does a slow process miss more in the UTCL1 / stall
# more at the L2's memory side than a fast one?      usage: tools/c4_diag.sh <tag> [processes-per-group]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-c4diag}; N=${2:-4}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
G1="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum"
G2="TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum"
G3="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"
for g in 1 2 3; do
  eval grp=\$G$g
  for k in $(seq 1 $N); do
    d=$O/g${g}_p$k
    timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o x -- \
      python $R/bench.py --agents 64 --batch 4096 --mode eager --protocol resident --steps 40 --warmup 5 --repeats 1 --region-ms 1 \
      --no-cpu-baseline --no-extra > $d.json 2> $d.err
  done
done
python - <<PY
import csv, glob, collections, json
import numpy as np
for g in (1, 2, 3):
    for f in sorted(glob.glob("$O/g%d_p*/x_counter_collection.csv" % g)):
        agg = collections.defaultdict(list); dur = {}
        for r in csv.DictReader(open(f)):
            if "k_duo" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        d = np.array(list(dur.values()))
        try:
            k_us = json.loads(open(f.replace("/x_counter_collection.csv", ".json")).read().strip().splitlines()[-1])["roofline"]["kernel_us_per_launch"]
        except Exception:
            k_us = float("nan")
        print("group %d %s: k_duo n=%d duration median %.2f us (p10 %.2f p90 %.2f) | bench k_us %.2f | " % (g, f.split("/")[-2], len(d), np.median(d), np.percentile(d, 10), np.percentile(d, 90), k_us)
              + "  ".join("%s %.4g" % (k.replace("_sum", ""), np.mean(v)) for k, v in sorted(agg.items())))
PY
