#!/usr/bin/env python3
"""The headline step kernel with the moves as fp32 one-hot rows [A][B][5] vs int32 ids [A][B]: HIP-event slope per launch."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import multiagent_particle_envs_amd as mpe  # noqa: E402

torch.cuda.set_device(0)
for B in (65536, 16384, 1048576):
    leg = bench.Leg(mpe, "simple_spread", 3, B, 25, 0, 1, 0)
    n = 400 if B <= 65536 else 50
    a = leg.kernel_time_us(torch, "graph", n=n, protocol="resident")
    b = leg.kernel_time_us(torch, "graph", n=n, protocol="resident_ids")
    a2 = leg.kernel_time_us(torch, "graph", n=n, protocol="resident")
    print("spread N=3 B=%d: one-hot rows %.3f us, ids %.3f us, rows again %.3f us" % (B, a, b, a2), flush=True)
    leg.release()
    torch.cuda.empty_cache()
