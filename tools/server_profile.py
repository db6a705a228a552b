#!/usr/bin/env python3
"""The step server's launch under rocprofv3 (kernel trace / PMC passes): N launches of T commanded steps each, with every
command issued BEFORE the launch -- a ring of T move tensors drawn ahead, the doorbell rung T ahead -- so that the launch never
waits and the profile still works when the profiler serialises dispatches (PMC collection does: a doorbell launch could not
overtake the resident server there).  Same kernel, same per-step work as the timed protocol (fresh moves read from HBM every
step, in-launch resets every 25 steps, every step's rows / rewards / dones / state written through).

    rocprofv3 --kernel-trace --stats ... -- python tools/server_profile.py [worlds] [T] [launches] [scenario]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import _abi  # noqa: E402
from multiagent_particle_envs_amd.rollout import StepServer  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    env = mpe.make_env(sys.argv[4] if len(sys.argv) > 4 else "simple_spread", batch_size=B, seed=0)
    A = env.n
    moves = torch.empty((T, A, B, _abi.MPE_ACTION_DIM), dtype=torch.float32, device="cuda")
    L = _abi.lib()
    srv = StepServer(env, moves, slots=2, episode_len=25, timeout_s=3.0, probe=False)
    for k in range(N):
        _abi.check(L.mpe_random_actions_block(moves.data_ptr(), None, A, B, 0, k * T, T, 0, _abi.raw_stream(env.world.device)), "draw")
        torch.cuda.synchronize()
        srv.served_to += T          # (commands first: ring() checks them against the launches started so far)
        srv.ring(T)
        srv.served_to -= T
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        srv.launch_events = []
        srv.start(T)
        srv.join()
        torch.cuda.synchronize()
        srv.check()
        u = srv.launch_events[0][0].elapsed_time(srv.launch_events[0][1]) * 1e3
        print("launch %d: %d steps, %.1f us = %.3f us per step (HIP events on the server's stream)" % (k, T, u, u / T), flush=True)


if __name__ == "__main__":
    main()
