#!/usr/bin/env python3
"""Regenerate profiles/pmc_traffic.json from the committed PMC summaries it cites.

    python profiles/make_pmc_traffic.py            # rewrite pmc_traffic.json
    python profiles/make_pmc_traffic.py --check    # exit 1 if the committed JSON differs from what the files say

Every entry of pmc_traffic.json (what bench.py prints as `roofline.traffic`) is the LAST line of the file it names --
`traffic_bytes_per_launch T   (reads R + writes W)`, written by profiles/pmc_summary.py from the rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md) -- and nothing else: no hand-typed number.
SOURCES maps bench.py's key (scenario_A<agents>_L<landmarks>_B<worlds>) to the newest committed summary of that launch.
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

# key -> (file under profiles/, the bench.py command the passes profiled)
SOURCES = {
    "simple_spread_A3_L3_B65536": ("r5_pmc_spread3_B65536.txt", "bench.py --mode eager --protocol resident"),
    "simple_spread_A3_L3_B1048576": ("r4_pmc_spread3_B1M.txt", "bench.py --mode eager --protocol resident --batch 1048576"),
    "simple_tag_A4_L2_B16384": ("r4_pmc_tag_B16384.txt", "bench.py --mode eager --protocol resident --scenario simple_tag --batch 16384"),
    "simple_spread_A64_L64_B4096": ("r4_pmc_spread64_B4096.txt", "bench.py --mode eager --protocol resident --agents 64 --batch 4096"),
    "simple_spread_A3_L3_B4096": ("r4_pmc_spread3_B4096.txt", "bench.py --mode eager --protocol resident --batch 4096"),
    # the step server's launch (mpe::k_split<..., SERVE>): ONE launch = 1000 commanded steps (tools/server_profile.py)
    "served_simple_spread_A3_L3_B65536": ("r6_pmc_served_spread3_B65536.txt", "tools/server_profile.py 65536 1000 4", 1000),
}
# key -> the newest committed `rocprofv3 --kernel-trace --stats` summary (tools/trace_summary.py) of that launch: bench.py prints
# the dominant kernel's mean / median duration from it as `roofline.kernel_us_rocprof` beside its own HIP-event slope
TRACES = {
    "simple_spread_A3_L3_B65536": "r6_spread3_B65536_kernel_trace_summary.txt",
    "simple_spread_A3_L3_B1048576": "r6_spread3_B1M_kernel_trace_summary.txt",
    "simple_tag_A4_L2_B16384": "r5_tag_B16384_kernel_trace_summary.txt",
    "simple_spread_A64_L64_B4096": "r5_spread64_B4096_kernel_trace_summary.txt",
    "simple_spread_A3_L3_B4096": "r5_spread3_B4096_kernel_trace_summary.txt",
    "served_simple_spread_A3_L3_B65536": "r6_served_spread3_B65536_kernel_trace_summary.txt",
}
ROW = re.compile(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)%\s*$")
LAST = re.compile(r"^traffic_bytes_per_launch\s+(\d+)\s+\(reads\s+(\d+)\s+\+\s+writes\s+(\d+)\)\s*$")


def entry(fname, command):
    with open(os.path.join(HERE, fname)) as f:
        lines = [l.rstrip("\n") for l in f if l.strip()]
    m = LAST.match(lines[-1])
    if not m:
        raise SystemExit("%s: last line is not a traffic_bytes_per_launch line: %r" % (fname, lines[-1]))
    kernel = next((l.split(None, 2)[2] for l in lines if l.startswith("# Kernel_Name")), None)
    t, r, w = (int(x) for x in m.groups())
    assert t == r + w or abs(t - r - w) <= 1, (fname, t, r, w)
    return {"traffic_bytes_per_launch": t, "read_bytes": r, "write_bytes": w, "kernel": kernel,
            "source": "profiles/%s (last line; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `%s`, summarised by "
                      "profiles/pmc_summary.py: KiB units, FETCH_SIZE x2 per MI355X_MICROARCH.md)" % (fname, command)}


def trace_entry(fname):
    """Mean / median duration of the step kernel (the first `mpe::` row = the one with the largest share) in a kernel-trace summary."""
    with open(os.path.join(HERE, fname)) as f:
        for l in f:
            m = ROW.match(l.rstrip("\n"))
            if m and not l.startswith("#") and "mpe::" in m.group(1):
                return {"mean_us": float(m.group(6)), "median_us": float(m.group(3)), "calls": int(m.group(2)),
                        "source": "profiles/%s" % fname}
    raise SystemExit("%s: no mpe:: kernel row" % fname)


def build():
    out = {}
    for k, v in SOURCES.items():
        out[k] = entry(v[0], v[1])
        if len(v) > 2:      # a launch of several steps: per-step figures are the launch's / steps_per_launch
            out[k]["steps_per_launch"] = v[2]
    for k, f in TRACES.items():
        out[k]["kernel_trace"] = trace_entry(f)
        if "steps_per_launch" in out[k]:
            out[k]["kernel_trace"]["steps_per_launch"] = out[k]["steps_per_launch"]
    return out


def main():
    new = build()
    path = os.path.join(HERE, "pmc_traffic.json")
    if "--check" in sys.argv:
        old = json.load(open(path))
        bad = [k for k in new if old.get(k) != new[k]] + [k for k in old if k not in new]
        if bad:
            print("pmc_traffic.json differs from the cited files for: %s" % ", ".join(sorted(set(bad))))
            sys.exit(1)
        print("pmc_traffic.json == the cited files' last lines (%d entries)" % len(new))
        return
    with open(path, "w") as f:
        json.dump(new, f, indent=1)
        f.write("\n")
    for k, v in new.items():
        print("%-36s %14d B   %s" % (k, v["traffic_bytes_per_launch"], SOURCES[k][0]))


if __name__ == "__main__":
    main()
