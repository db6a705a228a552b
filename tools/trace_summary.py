#!/usr/bin/env python3
"""Per-dispatch summary of a rocprofv3 --kernel-trace CSV (x_kernel_trace.csv): for every kernel the number of
dispatches and the distribution of its DURATION (end - start: median, p10, p90, mean, min, max), and for the dominant
kernel also its PERIOD -- start-to-start time of consecutive dispatches with nothing else between them -- which is what
`bench.py`'s `kernel_us_per_launch` measures with HIP events (time per dependent launch, launch gap included).

    python tools/trace_summary.py gpurun_out/.../x_kernel_trace.csv "command that was profiled" > profiles/NAME.txt

Why both: under the profiler every dispatch is bracketed by timestamp packets, so a 3 us kernel's recorded duration
has a long right tail (profiler serialisation, the first dispatches after a graph launch) and its mean says little;
the median and the period are the figures to compare with the bench line.
"""
import csv
import sys

import numpy as np


def main():
    path, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r[1])
    names = {}
    for n, s, e in rows:
        names.setdefault(n, []).append(e - s)
    tot = sum(sum(v) for v in names.values()) or 1
    print("# rocprofv3 --kernel-trace, per-dispatch summary (tools/trace_summary.py) of %s" % path.split("/")[-1])
    if cmd:
        print("# command: %s" % cmd)
    print("# %-90s %8s %9s %9s %9s %9s %9s %9s %6s" % ("kernel", "calls", "med_us", "p10_us", "p90_us", "mean_us", "min_us", "max_us", "pct"))
    order = sorted(names, key=lambda n: -sum(names[n]))
    for n in order:
        d = np.array(names[n], dtype=np.float64) / 1e3
        print("%-92s %8d %9.3f %9.3f %9.3f %9.3f %9.3f %9.3f %5.1f%%" % (n[:92], len(d), np.median(d), np.percentile(d, 10), np.percentile(d, 90),
                                                                          d.mean(), d.min(), d.max(), 100.0 * d.sum() * 1e3 / tot))
    top = order[0]
    per = [rows[k + 1][1] - rows[k][1] for k in range(len(rows) - 1) if rows[k][0] == top and rows[k + 1][0] == top]
    if per:
        p = np.array(per, dtype=np.float64) / 1e3
        p = p[p < 20 * np.median(p)]          # gaps between graph launches / repeats are not periods
        print("# dominant kernel: %s" % top[:110])
        print("# period (start-to-start of back-to-back dispatches, n=%d): median %.3f us  p10 %.3f  p90 %.3f  mean %.3f" %
              (len(p), np.median(p), np.percentile(p, 10), np.percentile(p, 90), p.mean()))
        d = np.array(names[top], dtype=np.float64) / 1e3
        print("# duration: median %.3f us; share of dispatches above 1.5 x median: %.2f%% (the mean's right tail)" %
              (np.median(d), 100.0 * (d > 1.5 * np.median(d)).mean()))


if __name__ == "__main__":
    main()
