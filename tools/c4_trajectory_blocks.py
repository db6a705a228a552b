#!/usr/bin/env python3
"""C4's rollout (simple_spread N=64, B=4096, 25 steps per launch) writes a 25 x 403 MB trajectory: 73-74 us per step where the
per-step launch on a fast buffer takes 67.  Is that a mix of fast and slow blocks?  Times the STEP kernel on each of the 25
trajectory blocks in turn (same state, same moves), then the rollout kernel itself.

    python tools/c4_trajectory_blocks.py > profiles/r4_c4_trajectory_blocks.txt
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import _abi  # noqa: E402
from multiagent_particle_envs_amd.rollout import RandomRollout, Trajectory  # noqa: E402


def main():
    N, B, T = 64, 4096, 25
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N, seed=0)
    rr = RandomRollout(env, episode_len=T, pool=4, regenerate=False)
    traj = Trajectory(env, T)
    L, st = _abi.lib(), _abi.raw_stream(env.world.device)
    bufs = env._sets[0].bufs
    desc = rr._desc
    per = traj.obs_flat.numel() // T
    print("# simple_spread N=%d, %d worlds: one trajectory of %d blocks x %.1f MB (one allocation at 0x%x); placement probe of the env's own "
          "buffers: %s" % (N, B, T, per * 4 / 1e6, traj.obs_flat.data_ptr(), env.placement_probe))

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
    own = [time_on(env._sets[k].obs.data_ptr()) for k in range(2)]
    print("step kernel on the env's own (probed) output sets: %.2f / %.2f us" % tuple(own))
    times = [time_on(traj.obs_flat.data_ptr() + t * per * 4) for t in range(T)]
    print("step kernel on trajectory block t (us): " + " ".join("%.1f" % x for x in times))
    fast = sum(1 for x in times if x < 1.08 * min(times))
    print("  min %.2f  median %.2f  max %.2f  mean %.2f   blocks within 8 %% of the fastest: %d of %d" %
          (min(times), sorted(times)[T // 2], max(times), sum(times) / T, fast, T))
    for _ in range(2):
        rr.fused(T, traj)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        rr.fused(T, traj)
    e1.record()
    torch.cuda.synchronize()
    print("fused rollout (k_duo_roll, %d steps per launch, every step into its block): %.2f us per step   (mean of the per-block step times: %.2f)"
          % (T, e0.elapsed_time(e1) * 1e3 / (6 * T), sum(times) / T))


if __name__ == "__main__":
    main()
