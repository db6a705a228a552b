#!/usr/bin/env python3
"""C4 follow-up (round 4, second and last VMM session): is a buffer slow BECAUSE its physical chunks are consecutive?

tools/vmm_c4.py found (box 34): every 64 MiB physical chunk alone takes the same time (11.2 us +- 1.5 %, both write patterns),
and of six 384 MiB buffers composed from 36 chunks the ONLY slow one (83 us vs 67-70) was chunks 0..5 in creation order.
Here: consecutive runs of chunks against strided / shuffled / rotated orders of the same pool, at 64, 8 and 2 MiB chunk sizes.

    python tools/vmm_c4_order.py > profiles/r4_c4_vmm_order_box<k>.txt
"""
import ctypes as C
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import _abi  # noqa: E402
from multiagent_particle_envs_amd.rollout import RandomRollout  # noqa: E402
from vmm_c4 import DevBuf  # noqa: E402


def main():
    vmm = C.CDLL(os.path.join(ROOT, "tools", "vmm", "libvmm.so"))
    vmm.vmm_create.restype, vmm.vmm_create.argtypes = C.c_longlong, [C.c_int, C.c_longlong]
    vmm.vmm_compose.restype, vmm.vmm_compose.argtypes = C.c_void_p, [C.POINTER(C.c_int), C.c_int]
    vmm.vmm_last_error.restype = C.c_char_p
    N, B = 64, 4096
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N, seed=0, probe_placement=False)
    rr = RandomRollout(env, episode_len=0, pool=2, regenerate=False)
    L, st = _abi.lib(), _abi.raw_stream(env.world.device)
    bufs, desc = env._sets[0].bufs, rr._desc
    nfl = env._sets[0].obs.numel()
    total = nfl * 4

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

    tb = [torch.empty(nfl, dtype=torch.float32, device="cuda") for _ in range(4)]
    print("torch allocations (4 x 384 MiB): step %s us" % " ".join("%.1f" % timed(step_on(t.data_ptr())) for t in tb))
    del tb
    torch.cuda.empty_cache()
    rnd = random.Random(7)
    for mib, pool in ((64, 30), (8, 240), (2, 960)):
        chunk = vmm.vmm_create(pool, mib << 20)
        if chunk != mib << 20:
            print("vmm_create(%d MiB) -> %d: %s" % (mib, chunk, vmm.vmm_last_error().decode()))
            vmm.vmm_destroy()
            continue
        per = total // chunk
        print("# %d physical chunks of %d MiB; a buffer = %d chunks" % (pool, mib, per))

        def run(ks, label):
            arr = (C.c_int * len(ks))(*ks)
            p = vmm.vmm_compose(arr, len(ks))
            if not p:
                print(label, "compose failed:", vmm.vmm_last_error().decode())
                return
            t = torch.as_tensor(DevBuf(p, nfl), device="cuda")
            t.zero_()
            torch.cuda.synchronize()
            print("%-44s step %6.1f us" % (label, timed(step_on(t.data_ptr()))), flush=True)
        for r in range(pool // per):
            run(list(range(r * per, (r + 1) * per)), "consecutive chunks %d..%d" % (r * per, (r + 1) * per - 1))
        run(list(range(per - 1, -1, -1)), "chunks %d..0 (reversed)" % (per - 1))
        run(list(range(per // 2, per)) + list(range(0, per // 2)), "first run rotated by half")
        if pool >= 2 * per:
            run(list(range(0, 2 * per, 2)), "every 2nd chunk of 0..%d" % (2 * per - 1))
            run([k for pair in zip(range(0, per // 2), range(per, per + per // 2)) for k in pair] if per % 2 == 0 else list(range(per)),
                "two runs interleaved chunk by chunk")
        for trial in range(4):
            ks = rnd.sample(range(pool), per)
            run(ks, "random %d of %d (trial %d)" % (per, pool, trial))
        ks = list(range(per))
        rnd.shuffle(ks)
        run(ks, "chunks 0..%d shuffled" % (per - 1))
        vmm.vmm_destroy()


if __name__ == "__main__":
    main()
