#!/usr/bin/env python3
"""Is the fast / slow split between 402.7 MB observation buffers (tools/c4_placement.py) a property of the WRITE PATTERN?
For each of K separately allocated buffers: a linear sweep (mode 0) and the row stream of the C4 step kernel (64 agent
blocks at 6 MiB strides, pieces of wpw rows) at several piece sizes, HIP-event time per launch; the product step kernel
on the same buffer beside them.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/wp/libwp.so tools/wp/write_pattern.hip
    python tools/write_pattern.py [K]
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
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    wp = C.CDLL(os.path.join(ROOT, "tools", "wp", "libwp.so"))
    wp.wp_launch.argtypes = [C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
    N, B, D = 64, 4096, 384
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N, seed=0, probe_placement=False)
    rr = RandomRollout(env, episode_len=0, pool=2, regenerate=False)
    L = _abi.lib()
    st = _abi.raw_stream(env.world.device)
    bufs = env._sets[0].bufs
    nfl = env._sets[0].obs.numel()
    desc = rr._desc

    def timed(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    def step_on(ptr):
        bufs.obs = ptr
        bufs.act, bufs.ids, bufs.u = rr.pool[0].data_ptr(), None, None
        return lambda: L.mpe_step(C.byref(desc), C.byref(bufs), B, st)
    modes = [("rows4 sc1", 33, 4, 0, 0), ("rows4 nt", 17, 4, 0, 0), ("tr 4K", 3, 4, 0, 0)] + \
            [("lock%d sc1" % k, 36, 4, k, 0) for k in (1, 2, 4, 8, 16)] + [("lock%d nt" % k, 20, 4, k, 0) for k in (1, 2, 4)] + \
            [("lock1 w8", 36, 8, 1, 0), ("lock2 w8", 36, 8, 2, 0)]
    print("%-4s %9s %9s %9s | %s" % ("buf", "mpe_step", "fill_(0)", "fill_(x)", " | ".join("%-9s" % m[0][-9:] for m in modes)), flush=True)
    keep = []
    for k in range(K):
        t = torch.empty(nfl, dtype=torch.float32, device=env.world.device)
        keep.append(t)
        row = [timed(step_on(t.data_ptr())), timed(lambda: t.fill_(0.0)), timed(lambda: t.fill_(1.2345))]
        for _, mode, wpw, rot, grid in modes:
            row.append(timed(lambda: wp.wp_launch(mode, t.data_ptr(), N, B, D, wpw, rot, grid, st)))
        print("%-4d %9.1f %9.1f %9.1f | %s" % (k, row[0], row[1], row[2], " | ".join("%-9.1f" % x for x in row[3:])), flush=True)


if __name__ == "__main__":
    main()
