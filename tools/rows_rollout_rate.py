#!/usr/bin/env python3
"""Fused T-step rollouts of row-program envs (mpe_rollout_rows) against the same steps as per-step launches, 65 536 worlds:
us per env step, interpreted and compiled in.    python tools/rows_rollout_rate.py > profiles/r4_rows_rollout_rate.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd.rollout import RandomRollout, Trajectory  # noqa: E402
import test_rowspec as tr  # noqa: E402


def timed(fn, steps, n=8):
    for _ in range(2):
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
        t = e0.elapsed_time(e1) * 1e3 / (n * steps)
        best = t if best is None else min(best, t)
    return best


def main():
    B, T = 65536, 25
    print("# us per env step at %d worlds, episodes of %d steps: per-step launches (a HIP graph of mpe_step_rows, fresh block-drawn moves," % (B, T))
    print("# a reset per episode) vs ONE mpe_rollout_rows launch per episode (state in LDS, moves and resets drawn in the kernel), with and")
    print("# without a trajectory (every step's rows kept / only the last step's)")
    cases = [("corral", lambda c: tr.corral_env(B, compile_program=c)),
             ("simple_spread as a program", lambda c: tr.make_spec_env("simple_spread", B, compile_program=c)),
             ("simple_adversary(4,2)", lambda c: mpe.make_env("simple_adversary", batch_size=B, num_agents=4, num_adversaries=2, compile_program=c)),
             ("simple_adversary(6,2)", lambda c: mpe.make_env("simple_adversary", batch_size=B, num_agents=6, num_adversaries=2, compile_program=c)),
             ("simple_world_comm(2,3)", lambda c: mpe.make_env("simple_world_comm", batch_size=B, num_good_agents=2, num_adversaries=3, compile_program=c))]
    for name, make in cases:
        for compiled in (False, True):
            env = make(False)
            if compiled:
                assert env.compile_program()
            rr = RandomRollout(env, episode_len=T, pool=T, regenerate=True)
            g = rr.capture(2 * T)
            per_step = timed(g.replay, 2 * T)
            traj = Trajectory(env, T)
            fused_traj = timed(lambda: rr.fused(T, traj), T)
            fused_last = timed(lambda: rr.fused(T), T)
            print("%-28s %-12s per-step launches %6.2f | fused rollout, trajectory kept %6.2f | last step only %6.2f   (%d ops)"
                  % (name, "compiled in" if compiled else "interpreted", per_step, fused_traj, fused_last, env._prog.n_ops))
            del rr, g, traj, env
            torch.cuda.empty_cache()
    # episodes that end by the program's done tests / the horizon, per world, inside the launch (mpe_rollout_rows_episode)
    for compiled in (False, True):
        env = tr.corral_env(B, arena=0.95, max_episode_steps=T, auto_reset=True)
        if compiled:
            assert env.compile_program()
        env.reset()
        rr = RandomRollout(env, episode_len=0, pool=T, regenerate=True)
        g = rr.capture(2 * T)
        per_step = timed(g.replay, 2 * T)
        traj = Trajectory(env, T)
        fused_traj = timed(lambda: rr.fused(T, traj), T)
        print("%-28s %-12s per-step launches %6.2f | fused rollout, trajectory kept %6.2f |   (done_spec + horizon %d + auto_reset: the episodes"
              " end inside the launches)" % ("corral with an arena", "compiled in" if compiled else "interpreted", per_step, fused_traj, T))
        del rr, g, traj, env
        torch.cuda.empty_cache()
    env = mpe.make_env("simple_spread", batch_size=B)
    rr = RandomRollout(env, episode_len=T, pool=T, regenerate=True)
    g = rr.capture(2 * T)
    traj = Trajectory(env, T)
    print("%-28s %-12s per-step launches %6.2f | fused rollout, trajectory kept %6.2f | last step only %6.2f"
          % ("simple_spread", "fused kernel", timed(g.replay, 2 * T), timed(lambda: rr.fused(T, traj), T), timed(lambda: rr.fused(T), T)))


if __name__ == "__main__":
    main()
