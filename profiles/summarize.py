#!/usr/bin/env python3
"""Turn a rocprofv3 results database (ROCm 7.2 writes rocpd sqlite by default) into the plain-text
kernel summary kept under profiles/ (same columns as `rocprofv3 --stats`' kernel_stats.csv).

    python profiles/summarize.py gpurun_out/prof_dir/x_results.db "command line that was profiled" > profiles/NAME.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cmd = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = db.cursor()
    print("# rocprofv3 --kernel-trace --stats summary (from %s)" % sys.argv[1].split("/")[-1])
    if cmd:
        print("# command: %s" % cmd)
    print("# %-100s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    rows = list(cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
        "group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    for name, n, s, a, mn, mx in rows:
        print("%-102s %8d %14.3f %12.3f %12.3f %12.3f %6.2f%%" % (name[:102], n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    r = cur.execute("select vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x from kernels "
                    "where name = ? limit 1", (rows[0][0],)).fetchone()
    print("# top kernel resources: vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s workgroup=%s grid=%s" % r)


if __name__ == "__main__":
    main()
