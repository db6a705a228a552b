#!/usr/bin/env python3
"""C4 placement: is the FIRST large allocation of a process the fast one, and does that survive a larger block?
    python tools/c4_first.py single|double|triple     (one process each; tools/sessions/r3_session12.sh interleaves them)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import _abi  # noqa: E402
from multiagent_particle_envs_amd.rollout import RandomRollout  # noqa: E402


def main():
    mode = sys.argv[1]
    N, B = 64, 4096
    nfl = N * (6 * N) * B
    mult = {"single": 1, "double": 2, "triple": 3}[mode]
    dev = torch.device("cuda", 0)
    torch.cuda.init()
    block = torch.zeros(mult * nfl, dtype=torch.float32, device=dev)       # the first large allocation of the process
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N, seed=0, probe_placement=False)
    rr = RandomRollout(env, episode_len=0, pool=4, regenerate=False)
    L, st, bufs, desc = _abi.lib(), _abi.raw_stream(dev), env._sets[0].bufs, rr._desc

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
    parts = [time_on(block[k * nfl:(k + 1) * nfl].data_ptr()) for k in range(mult)]
    later = [time_on(s.obs.data_ptr()) for s in env._sets]
    print("%-7s first allocation (%4d MiB): %s us | the env's own two sets (allocated later): %s us" %
          (mode, mult * 384, " ".join("%.1f" % x for x in parts), " ".join("%.1f" % x for x in later)), flush=True)


if __name__ == "__main__":
    main()
