#!/usr/bin/env python3
"""Where does the N=64 step go?  Times (HIP events, graph replay) the full fused step, the physics
half alone (mpe_world_step), the output half alone (mpe_observe), and plain fills / copies of the
observation buffer (the write-bandwidth reference for this access size)."""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _abi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = mpe.make_env("simple_spread", batch_size=B, num_agents=N)
env._ensure_buffers()
A = N
act = torch.zeros((A, B, 5), device="cuda")
act[..., 1] = 1.0
L = _abi.lib()
out = env._sets[0]
b = out.bufs
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
gen = env.world.scenario_desc(_abi.MPE_SCN_GENERIC)
gb = _abi.MpeBuffers()
gb.pos, gb.vel, gb.act = env.world.pos.data_ptr(), env.world.vel.data_ptr(), act.data_ptr()
gb.entity_table = env._entity_table.data_ptr()


def full():
    b.act, b.ids, b.u = act.data_ptr(), None, None
    _abi.check(L.mpe_step(C.byref(env._desc), C.byref(b), B, st()))


def phys():
    _abi.check(L.mpe_world_step(C.byref(gen), C.byref(gb), B, st()))


def outp():
    b.act = b.ids = b.u = None
    _abi.check(L.mpe_observe(C.byref(env._desc), C.byref(b), B, st()))


other = torch.empty_like(out.obs)


def fill():
    out.obs.zero_()


def copy():
    other.copy_(out.obs)


def timeit(fn, n=50):
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


mb = out.obs.numel() * 4 / 1e6
def obs_only():
    b.act = b.ids = b.u = None
    keep = (b.rew, b.done)
    b.rew = b.done = None
    _abi.check(L.mpe_observe(C.byref(env._desc), C.byref(b), B, st()))
    b.rew, b.done = keep


for name, fn in (("full step", full), ("obs no reward", obs_only), ("physics only", phys), ("observe only", outp), ("obs fill", fill), ("obs copy", copy)):
    us = timeit(fn)
    print("%-14s %9.2f us   (obs %.0f MB -> %.0f GB/s if it were all obs traffic)" % (name, us, mb, mb / us * 1e3 / 1e3 * 1e0))
