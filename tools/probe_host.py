#!/usr/bin/env python3
"""Host cost of one MultiAgentEnv.step() call: tiny batch (the kernel is ~2 us), many calls, cProfile."""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multiagent_particle_envs_amd as mpe

env = mpe.make_env("simple_spread", batch_size=256)
act = torch.zeros((3, 256, 5), device="cuda")
act[..., 1] = 1
env.reset()
for _ in range(1000):
    env.step(act)
torch.cuda.synchronize()
n = 20000
t0 = time.perf_counter()
for _ in range(n):
    env.step(act)
torch.cuda.synchronize()
print("step(): %.2f us per call (host-bound, B=256)" % ((time.perf_counter() - t0) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(5000):
    env.step(act)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
