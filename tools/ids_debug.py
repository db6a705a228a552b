#!/usr/bin/env python3
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import sharding  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
rv = sharding.Rendezvous(0, 1, dev, "auto")
B = 65536
leg = bench.Leg(mpe, "simple_spread", 3, B, 25, 0, 1, 0)
print("fresh leg: ids %.3f rows %.3f" % (leg.kernel_time_us(torch, "graph", protocol="resident_ids"), leg.kernel_time_us(torch, "graph", protocol="resident")), flush=True)
dt, R, _, _ = leg.timed(torch, rv, dev, "graph", "fresh", 20, 5, 2, 300)
print("after timed(fresh): ids %.3f rows %.3f" % (leg.kernel_time_us(torch, "graph", protocol="resident_ids"), leg.kernel_time_us(torch, "graph", protocol="resident")), flush=True)
dt, R, _, _ = leg.timed(torch, rv, dev, "graph", "fresh_ids", 20, 5, 2, 300)
k1 = leg.kernel_time_us(torch, "graph", protocol="resident_ids")
print("after timed(fresh_ids): ids %.3f  timing %s" % (k1, leg.last_kernel_timing), flush=True)
k2 = leg.kernel_time_us(torch, "graph", protocol="resident")
print("  rows %.3f timing %s" % (k2, leg.last_kernel_timing), flush=True)
w = leg.env.world
print("  |pos| max %.3g, finite %s" % (float(w.pos.abs().max()), bool(torch.isfinite(w.pos).all())))
leg.rolls.pop("fresh_ids", None); leg.rolls.pop("resident_ids", None)
print("after popping the ids rolls: ids %.3f rows %.3f" % (leg.kernel_time_us(torch, "graph", protocol="resident_ids"), leg.kernel_time_us(torch, "graph", protocol="resident")), flush=True)
