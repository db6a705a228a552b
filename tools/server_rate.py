#!/usr/bin/env python3
"""The step server against the launched steps, same box, same process order: for each config the headline protocol's ms per step
as launches (bench.Leg graph protocol: fresh moves, reset every 25) and as commanded steps (bench.served_leg).

    python tools/server_rate.py [C2 C3 C5 ...]      # or scenario:agents:worlds
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import sharding  # noqa: E402

NAMED = {"C2": ("simple_spread", 3, 4096), "C3": ("simple_tag", 3, 16384), "C5": ("simple_spread", 3, 65536),
         "S16": ("simple_spread", 3, 16384), "S32": ("simple_spread", 3, 32768)}


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rv = sharding.Rendezvous(0, 1, dev)
    for spec in (sys.argv[1:] or ["C2", "C3", "C5"]):
        scn, ag, B = NAMED[spec] if spec in NAMED else (spec.split(":")[0], int(spec.split(":")[1]), int(spec.split(":")[2]))
        leg = bench.Leg(mpe, scn, ag, B, 25, 0, 1, 0)
        d, R, _, _ = leg.timed(torch, rv, dev, "graph", "fresh", 200, 10, 3, 300.0)
        k = leg.kernel_time_us(torch, "graph", n=400)
        launched = d * 1e3 / (200 * R)
        kw = dict(leg.kw)
        leg.release()
        torch.cuda.empty_cache()
        s = bench.served_leg(torch, mpe, scn, ag, B, 25, 0, 300.0, kw=kw)
        p = s.get("per_step_doorbells", {})
        print("%-6s %s A=%d B=%d | launched: %.3f us/step (kernel slope %.3f) | served: %.3f us/step = %.3f G env-steps/s, frac_timed %.3f, x%.2f "
              "| with a doorbell launch per step: %s us/step"
              % (spec, scn, ag, B, launched * 1e3, k, s["ms_per_step"] * 1e3, s["value"] / 1e9, s["frac_timed_region"],
                 launched / s["ms_per_step"], ("%.3f" % (p["ms_per_step"] * 1e3)) if "ms_per_step" in p else p.get("error")), flush=True)


if __name__ == "__main__":
    main()
