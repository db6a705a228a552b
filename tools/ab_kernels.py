#!/usr/bin/env python3
"""Same-box A/B of library variants at kernel granularity: for each (scenario, agents, worlds) the step kernel's time
per launch (HIP events over 400 graph-replayed dependent launches, best of 3) and the fused rollout's time per step.
One process per variant (MPE_HIP_LIB picks the library at import); tools/ab_matrix.sh interleaves variants and repeats.

    MPE_HIP_LIB=.../libmpe_hip_ab_x.so python tools/ab_kernels.py tag:3:16384 spread:3:4096 ...
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import multiagent_particle_envs_amd as mpe  # noqa: E402


def main():
    torch.cuda.set_device(0)
    tag = os.path.basename(os.environ.get("MPE_HIP_LIB", "base")).replace("libmpe_hip_ab_", "").replace(".so", "")
    out = []
    for spec in sys.argv[1:]:
        scn, ag, B = spec.split(":")[:3]      # optional 4th field: make_world arguments, "num_agents=6,num_adversaries=2"
        kw = {k: int(v) for k, v in (x.split("=") for x in spec.split(":")[3].split(","))} if spec.count(":") > 2 else None
        scn = {"tag": "simple_tag", "spread": "simple_spread"}.get(scn, scn)
        leg = bench.Leg(mpe, scn, int(ag), int(B), 25, 0, 1, 0, scenario_kw=kw)
        n = 400 if int(B) * leg.A < 400000 else 100
        k = leg.kernel_time_us(torch, "graph", n=n)
        f = leg.kernel_time_us(torch, "fused", n=n)
        out.append("%s step %.3f roll %.3f" % (spec, k, f))
        leg.release()
        torch.cuda.empty_cache()
    print("%-8s %s" % (tag, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
