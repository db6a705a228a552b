#!/usr/bin/env python3
"""C4 (simple_spread N=64): does the slow / fast split between observation buffers (tools/c4_placement.py) depend on the
STRIDE between the agents' row blocks?  At 4096 worlds agent i's rows start exactly 6 MiB after agent i-1's; a world
group's 64 row streams then differ by multiples of 6 MiB.  Other batch sizes give strides that are not a multiple of a
large power of two.  For each B: K separately allocated buffers, the step kernel timed on each, normalised to 4096 worlds.

    python tools/c4_stride.py [K] [B ...]
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import _abi  # noqa: E402
from multiagent_particle_envs_amd.rollout import RandomRollout  # noqa: E402


def run(B, K, N=64):
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N, seed=0, probe_placement=False)
    rr = RandomRollout(env, episode_len=0, pool=2, regenerate=False)
    L = _abi.lib()
    st = _abi.raw_stream(env.world.device)
    bufs = env._sets[0].bufs
    nfl = env._sets[0].obs.numel()
    desc = rr._desc

    def time_on(ptr, n=40):
        bufs.obs = ptr
        bufs.act, bufs.ids, bufs.u = rr.pool[0].data_ptr(), None, None
        for _ in range(4):
            L.mpe_step(C.byref(desc), C.byref(bufs), B, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            L.mpe_step(C.byref(desc), C.byref(bufs), B, st)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    keep, times = [], []
    for k in range(K):
        t = torch.empty(nfl, dtype=torch.float32, device=env.world.device)
        keep.append(t)
        times.append(time_on(t.data_ptr()) * 4096.0 / B)
    fill = torch.empty(nfl, dtype=torch.float32, device=env.world.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fill.fill_(0.0)
    e0.record()
    for _ in range(10):
        fill.fill_(0.0)
    e1.record()
    torch.cuda.synchronize()
    f_us = e0.elapsed_time(e1) * 100.0 * 4096.0 / B
    print("B %5d  agent stride %.4f MiB  fill %.1f us  step (us per 4096 worlds), %d buffers sorted: %s" %
          (B, nfl * 4 / N / (1 << 20), f_us, K, " ".join("%.1f" % x for x in sorted(times))), flush=True)
    del keep, fill, env, rr
    torch.cuda.empty_cache()


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    Bs = [int(x) for x in sys.argv[2:]] or [4096, 4160, 4032, 4352, 4096, 8192, 2048, 3072]
    for B in Bs:
        run(B, K)


if __name__ == "__main__":
    main()
