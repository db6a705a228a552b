#!/usr/bin/env python3
"""Where the cycles of one step go: run the MPE_PHASE_CLOCK build of k_split (tools/ab_build.sh clk split
-DMPE_PHASE_CLOCK) and print, per role, the mean shader-clock delta between the phase stamps of a step.

    MPE_HIP_LIB=.../libmpe_hip_ab_clk.so python tools/phase_clock.py simple_tag 16384 [roll|step]

agent wave stamps  0 step start | 1 World.step done | 2 published, at the barrier | 3 through the barrier |
                   4 siblings' positions read | 5 rows on their way
reward wave stamps 0 step start | 1 next moves drawn, at the barrier | 2 through the barrier | 3 rewards on their way
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd.rollout import RandomRollout, Trajectory  # noqa: E402


def main():
    scn, B, mode = sys.argv[1], int(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else "roll")
    kw = {"num_agents": int(sys.argv[4])} if len(sys.argv) > 4 else {}
    env = mpe.make_env(scn, batch_size=B, seed=0, **kw)
    rr = RandomRollout(env, episode_len=25, pool=25, regenerate=False)
    dbg = torch.zeros(4 * 16 * 32 * 8, dtype=torch.int64, device="cuda")
    A = len(env.world.agents)
    T = 25
    traj = Trajectory(env, T)
    for bufs in [s.bufs for s in env._sets] + [traj.bufs]:
        bufs.force = dbg.data_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        dbg.zero_()
        torch.cuda.synchronize()
        e0.record()
        if mode == "roll":
            rr.fused(T, traj)
        else:
            rr.enqueue(1)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    d = dbg.cpu().numpy().reshape(4, 16, 32, 8).astype(np.int64)
    nt = T if mode == "roll" else 1
    print("%s B=%d %s: launch %.2f us by events (%d steps)" % (scn, B, mode, ms * 1e3, nt))
    t_first = d[:, :, 0, 0][d[:, :, 0, 0] > 0].min()
    t_last = d[:, :, :nt, :].max()
    print("  first stamp -> last stamp: %d ticks; if the launch is ~that long, a tick is %.2f ns" % (t_last - t_first, ms * 1e6 / max(t_last - t_first, 1)))
    for role in range(16):
        st = d[:, role, :nt, :]
        if not st[:, :, 0].any():
            continue
        nst = 6 if st[:, :, 5].any() or st[:, :, 4].any() else 4
        steps = st[:, 1:, 0] - st[:, :-1, 0] if nt > 1 else None
        line = "  role %d (%s): " % (role, "agent" if nst == 6 else "reward")
        for k in range(1, nst):
            dk = st[:, (1 if nt > 1 else 0):, k] - st[:, (1 if nt > 1 else 0):, k - 1]
            line += "%d->%d %6.0f  " % (k - 1, k, dk.mean())
        if steps is not None:
            line += "| step-to-step %6.0f (min %d max %d)" % (steps.mean(), steps.min(), steps.max())
        else:
            line += "| entry->last stamp %6.0f" % (st[:, 0, nst - 1] - st[:, 0, 0]).mean()
        print(line)
    if nt == 1:   # the step kernel: when each wave reached its stamps relative to the workgroup's first stamp
        for blk in range(2):
            base = d[blk, :, 0, 0][d[blk, :, 0, 0] > 0].min()
            print("  workgroup %d, ticks after its first stamp:" % blk,
                  {r: [int(x - base) for x in d[blk, r, 0, :6] if x > 0] for r in range(A + 1) if d[blk, r, 0, 0] > 0})


if __name__ == "__main__":
    main()
