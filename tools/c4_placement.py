#!/usr/bin/env python3
"""C4 (simple_spread N=64, B=4096): is a launch's time a property of WHICH 403 MB observation buffer it writes?
(round-3 finding: under the ping-pong of the env's two output sets a process shows 69 / 85 / 69 / 85 us per dispatch.)
Times the step kernel on each of K separately allocated observation buffers (same state, same moves) and prints each
buffer's device address next to its time; then the same for sub-allocations at different offsets inside one big block.

    python tools/c4_placement.py [K] [brief [N B]]
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


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    brief = len(sys.argv) > 2 and sys.argv[2] == "brief"      # only the K separate allocations, one summary line
    N, B = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 4096)     # (any simple_spread shape)
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N, seed=0, probe_placement=False)
    rr = RandomRollout(env, episode_len=0, pool=4, regenerate=False)
    L = _abi.lib()
    st = _abi.raw_stream(env.world.device)
    bufs = env._sets[0].bufs
    nfl = env._sets[0].obs.numel()
    desc = rr._desc

    def time_on(ptr, n=60):
        bufs.obs = ptr
        bufs.act, bufs.ids, bufs.u = rr.pool[0].data_ptr(), None, None
        for _ in range(5):
            L.mpe_step(C.byref(desc), C.byref(bufs), B, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            L.mpe_step(C.byref(desc), C.byref(bufs), B, st)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    def show(tag, ptr, us):
        print("%-22s ptr 0x%012x  mod 2MiB %8d  mod 1GiB %5d MiB  bits[21:30] %s   %.2f us" %
              (tag, ptr, ptr % (2 << 20), (ptr % (1 << 30)) >> 20, format((ptr >> 21) & 0x1ff, "09b"), us), flush=True)
    print("obs buffer: %d floats = %.1f MB; agent block = %.3f MiB" % (nfl, nfl * 4 / 1e6, nfl * 4 / N / (1 << 20)))
    for k in range(2):
        show("env output set %d" % k, env._sets[k].obs.data_ptr(), time_on(env._sets[k].obs.data_ptr()))
    keep = []
    times = []
    for k in range(K):
        t = torch.empty(nfl, dtype=torch.float32, device=env.world.device)
        keep.append(t)
        times.append(time_on(t.data_ptr()))
        if not brief:
            show("separate alloc %d" % k, t.data_ptr(), times[-1])
    if brief:
        tag = os.path.basename(os.environ.get("MPE_HIP_LIB", "base")).replace("libmpe_hip_ab_", "").replace(".so", "")
        print("%-8s %d buffers, us per launch sorted: %s" % (tag, K, " ".join("%.1f" % x for x in sorted(times))), flush=True)
        return
    for k in range(2):   # the same buffers again: is the time a property of the buffer?
        show("again: alloc %d" % k, keep[k].data_ptr(), time_on(keep[k].data_ptr()))
    del keep
    torch.cuda.empty_cache()
    big = torch.empty(nfl + (256 << 20) // 4, dtype=torch.float32, device=env.world.device)   # +256 MiB of slack
    for off_mib in (0, 1, 2, 4, 8, 16, 32, 64, 128, 3, 6, 12):
        p = big.data_ptr() + (off_mib << 20)
        show("big + %d MiB" % off_mib, p, time_on(p))
    for off in (64, 256, 4096, 65536, 1 << 19):
        p = big.data_ptr() + off
        show("big + %d B" % off, p, time_on(p))


if __name__ == "__main__":
    main()
