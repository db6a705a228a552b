#!/usr/bin/env python3
"""Per-launch averages of rocprofv3 --pmc counter CSVs for one kernel (name substring).

    python profiles/pmc_summary.py gpurun_out/pmc_<tag> k_split > profiles/<name>.txt

Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory side;
on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (doubled here as the guide
prescribes; this kernel family's dword loads are full 256-byte wave accesses), WRITE_SIZE matched
the algorithmic store bytes to 0.2 % on the thread-per-world kernel and is used as is.
"""
import collections
import csv
import glob
import sys


def main():
    root, pat = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(list)
    meta = {}
    for f in sorted(glob.glob(root + "/*/x_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if (name.endswith(pat[4:]) or name.endswith(pat[4:] + ".kd")) if pat.startswith("end:") else pat in name:      # "end:_s": a suffix
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                          "SGPR_Count", "Scratch_Size")}
    print("# rocprofv3 --pmc per-launch averages, kernel matching %r under %s" % (pat, root))
    for k, v in meta.items():
        print("# %-16s %s" % (k, v[:110]))
    waves = None
    if "SQ_WAVES" in agg:
        waves = sum(agg["SQ_WAVES"]) / len(agg["SQ_WAVES"])
    for name in sorted(agg):
        v = agg[name]
        avg = sum(v) / len(v)
        extra = ""
        if name == "FETCH_SIZE":
            extra = "  KiB -> x2 (gfx950 correction) = %.3f MB HBM-side reads per launch" % (avg * 2 * 1024 / 1e6)
        elif name == "WRITE_SIZE":
            extra = "  KiB = %.3f MB HBM-side writes per launch" % (avg * 1024 / 1e6)
        elif waves and name.startswith("SQ_") and name != "SQ_WAVES":
            extra = "  (%.1f per wave)" % (avg / waves)
        print("%-24s n=%-5d avg=%16.1f%s" % (name, len(v), avg, extra))
    if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
        rd = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"]) * 2 * 1024
        wr = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"]) * 1024
        print("traffic_bytes_per_launch %.0f   (reads %.0f + writes %.0f)" % (rd + wr, rd, wr))


if __name__ == "__main__":
    main()
