#!/usr/bin/env python3
"""Step time over wall time: does the N=64 (store-bound) step settle into a faster or slower mode as the
GPU's clocks react to sustained load?  Replays a 30-launch graph back to back for a few seconds and
prints the per-step time of each ~0.25 s window."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _abi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
SEC = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
L = _abi.lib()
kw = {"num_agents": N} if N != 3 else {}
env = mpe.make_env("simple_spread", batch_size=B, **kw)
env._ensure_buffers()
act = torch.zeros((N, B, 5), device="cuda")
act[..., 1] = 1.0
b = env._sets[0].bufs
b.act, b.ids, b.u = act.data_ptr(), None, None
n = 30 if N > 6 else 300


def full():
    _abi.check(L.mpe_step(C.byref(env._desc), C.byref(b), B, C.c_void_p(torch.cuda.current_stream().cuda_stream)))


g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    full(); full()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n):
            full()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t_start = time.perf_counter()
while time.perf_counter() - t_start < SEC:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    reps = 0
    e0.record()
    while time.perf_counter() - t0 < 0.25:
        g.replay()
        reps += 1
        if reps % 8 == 0:
            torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    print("t=%5.2f s  %8.2f us/step" % (time.perf_counter() - t_start, e0.elapsed_time(e1) * 1e3 / (reps * n)))
