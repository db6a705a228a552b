#!/usr/bin/env python3
"""Is the N=64 step time a property of the process or of where its output buffer landed?  Builds several
envs in ONE process (all kept alive, so every obs block is a different allocation) and times the same
fused step on each (HIP-graph replay of 30 launches)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _abi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
L = _abi.lib()
envs = []
act = torch.zeros((N, B, 5), device="cuda")
act[..., 1] = 1.0


def timeit(fn, n=30):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


pad = []
for k in range(K):
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N)
    env._ensure_buffers()
    envs.append(env)
    out = env._sets[0]
    b = out.bufs
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def full(env=env, b=b):
        b.act, b.ids, b.u = act.data_ptr(), None, None
        _abi.check(L.mpe_step(C.byref(env._desc), C.byref(b), B, st()))
    us = timeit(full)
    print("env %d  obs @ 0x%x (mod 2MiB = 0x%06x)   full step %7.2f us" % (k, out.obs.data_ptr(), out.obs.data_ptr() % (2 << 20), us))
    pad.append(torch.empty((k + 1) * 12345 * 17, device="cuda"))   # shift the next allocation
