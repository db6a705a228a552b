#!/usr/bin/env python3
"""C4 (simple_spread N=64, 4096 worlds): can a fast observation buffer be CONSTRUCTED from chosen physical chunks?  (round 4, one bounded experiment)

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/vmm/libvmm.so tools/vmm/vmm_probe.hip
    python tools/vmm_c4.py [n_chunks=36] > profiles/r4_c4_vmm_box<k>.txt

1. N physical 64 MiB chunks through the HIP virtual-memory API (hipMemCreate); each is mapped alone and timed under two write
   patterns: persistent workgroups writing one contiguous slice each (the pattern that separated fast from slow 384 MiB
   allocations in round 3) and a fill (the placement-insensitive control).
2. The step kernel (`k_duo`) is timed on observation buffers of 6 chunks = 384 MiB each, composed (hipMemMap under one virtual
   range) from: the 6 fastest chunks, the 6 slowest, 6 in creation order -- next to ordinary torch allocations of the same size.
If per-chunk speed exists and composes, "fastest" beats every torch allocation of a box whose allocations are all slow.
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


class DevBuf(object):
    """A raw device range as something torch.as_tensor takes (CUDA array interface)."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def main():
    n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 36
    vmm = C.CDLL(os.path.join(ROOT, "tools", "vmm", "libvmm.so"))
    vmm.vmm_create.restype, vmm.vmm_create.argtypes = C.c_longlong, [C.c_int, C.c_longlong]
    vmm.vmm_probe.restype, vmm.vmm_probe.argtypes = C.c_double, [C.c_int, C.c_int, C.c_int, C.c_int]
    vmm.vmm_compose.restype, vmm.vmm_compose.argtypes = C.c_void_p, [C.POINTER(C.c_int), C.c_int]
    vmm.vmm_last_error.restype = C.c_char_p
    N, B = 64, 4096
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N, seed=0, probe_placement=False)
    rr = RandomRollout(env, episode_len=0, pool=2, regenerate=False)
    L, st = _abi.lib(), _abi.raw_stream(env.world.device)
    bufs, desc = env._sets[0].bufs, rr._desc
    nfl = env._sets[0].obs.numel()
    assert nfl * 4 == 6 * (64 << 20), nfl

    def timed(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / n
            best = t if best is None else min(best, t)
        return best

    def step_on(ptr):
        bufs.obs = ptr
        bufs.act, bufs.ids, bufs.u = rr.pool[0].data_ptr(), None, None
        return lambda: L.mpe_step(C.byref(desc), C.byref(bufs), B, st)

    print("# C4 step kernel (k_duo, 4096 worlds x 64 agents, 384 MiB of rows per launch) on observation buffers by provenance")
    torch_bufs = [torch.empty(nfl, dtype=torch.float32, device="cuda") for _ in range(6)]
    t_torch = [timed(step_on(t.data_ptr())) for t in torch_bufs]
    fills = [timed(lambda t=t: t.fill_(0.0)) for t in torch_bufs[:2]]
    print("torch allocations (6 x 384 MiB): step %s us   (fill %s us)" % (" ".join("%.1f" % x for x in t_torch), " ".join("%.1f" % x for x in fills)))
    chunk = vmm.vmm_create(n_chunks, 64 << 20)
    if chunk <= 0:
        print("vmm_create failed:", vmm.vmm_last_error().decode())
        return
    print("# %d physical chunks of %d MiB (hipMemCreate); per chunk, mapped alone: us per launch writing the whole chunk" % (n_chunks, chunk >> 20))
    slices = [vmm.vmm_probe(k, 0, 40, 2048) for k in range(n_chunks)]
    slices2 = [vmm.vmm_probe(k, 0, 40, 2048) for k in range(n_chunks)]
    fill = [vmm.vmm_probe(k, 1, 40, 0) for k in range(n_chunks)]
    print("%-5s %10s %10s %10s" % ("chunk", "slices", "slices(2)", "fill"))
    for k in range(n_chunks):
        print("%-5d %10.2f %10.2f %10.2f" % (k, slices[k], slices2[k], fill[k]))
    if min(slices) < 0:
        print("probe failed:", vmm.vmm_last_error().decode())
        return
    order = sorted(range(n_chunks), key=lambda k: slices[k] + slices2[k])
    spread = (slices[order[-1]] + slices2[order[-1]]) / (slices[order[0]] + slices2[order[0]])
    rep = sum(abs(a - b) for a, b in zip(slices, slices2)) / n_chunks
    print("# slowest / fastest chunk = %.3f; mean |pass 1 - pass 2| = %.2f us" % (spread, rep))

    def composed(ks, label):
        arr = (C.c_int * len(ks))(*ks)
        p = vmm.vmm_compose(arr, len(ks))
        if not p:
            print(label, "compose failed:", vmm.vmm_last_error().decode())
            return None
        t = torch.as_tensor(DevBuf(p, nfl), device="cuda")
        t.zero_()
        torch.cuda.synchronize()
        us = timed(step_on(t.data_ptr()))
        fl = timed(lambda: t.fill_(0.0))
        print("%-34s chunks %-28s step %.1f us   fill %.1f us" % (label, ks, us, fl))
        return us
    composed(order[:6], "6 fastest chunks")
    composed(order[6:12], "next 6 fastest")
    composed(order[-6:], "6 slowest chunks")
    composed(list(range(6)), "chunks 0-5 (creation order)")
    composed(order[:6][::-1], "6 fastest, reversed")
    composed([order[0], order[-1], order[1], order[-2], order[2], order[-3]], "fast / slow interleaved")
    t_again = [timed(step_on(t.data_ptr())) for t in torch_bufs[:3]]
    print("torch allocations again: step %s us" % " ".join("%.1f" % x for x in t_again))


if __name__ == "__main__":
    main()
