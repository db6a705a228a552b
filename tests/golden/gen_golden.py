#!/usr/bin/env python3
"""Record golden vectors from the UNMODIFIED reference (runs only in the build container).

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz

The reference tree at /root/reference is imported read-only (no bytecode is written into it)
through the shape-only `gym` stub in tests/golden/_gym_stub.  Nothing here is used at test time
on the GPU box: the tests read the committed .npz files.

What is recorded, per scenario (W worlds, T steps):
  seeds [W]           np.random.seed(seed) is called immediately before env.reset()
  pos0 [W,E,2] vel0   state after that reset (some worlds are then squeezed towards the origin so
                      that agents overlap and the contact path is exercised hard)
  act [T,W,A,5]       actions fed to env.step (one-hot, plus "soft" real-valued rows)
  obs{i} [T,W,D_i]    env.step outputs, per agent i      rew [T,W,A]   done [T,W,A]
  pos [T,W,E,2] vel [T,W,A,2]  world state after each step
  info_* [T,W,A]      benchmark_data tuples (make_env(..., benchmark=True))
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["SUPPRESS_MA_PROMPT"] = "1"
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "_gym_stub"), "/root/reference"]

import numpy as np  # noqa: E402
import warnings  # noqa: E402

warnings.filterwarnings("ignore")
from make_env import make_env  # noqa: E402  (the reference's factory, make_env.py:15)
from multiagent.environment import MultiAgentEnv  # noqa: E402
import multiagent.scenarios as ref_scenarios  # noqa: E402
from multiagent.core import World, Agent, Landmark  # noqa: E402


def spread_n(n):
    """simple_spread with N agents / N landmarks: only make_world's two hard-coded counts
    (simple_spread.py:11-12) change; every other method is the reference's own."""
    Base = ref_scenarios.load("simple_spread.py").Scenario

    class ScN(Base):
        def make_world(self):
            world = World()
            world.dim_c = 2
            world.collaborative = True
            world.agents = [Agent() for _ in range(n)]
            for i, agent in enumerate(world.agents):
                agent.name = "agent %d" % i
                agent.collide = True
                agent.silent = True
                agent.size = 0.15
            world.landmarks = [Landmark() for _ in range(n)]
            for i, landmark in enumerate(world.landmarks):
                landmark.name = "landmark %d" % i
                landmark.collide = False
                landmark.movable = False
            self.reset_world(world)
            return world

    sc = ScN()
    world = sc.make_world()
    return MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation, sc.benchmark_data)


def state_of(env, all_vel=False):
    ents = env.world.entities
    pos = np.array([e.state.p_pos for e in ents])
    vel = np.array([a.state.p_vel for a in (ents if all_vel else env.world.agents)])
    return pos, vel


def record(name, env, seeds, T, squeeze_every=0, squeeze=0.3, soft_every=4, ids=False, arng=None, all_vel=False):
    """all_vel: record the velocity of EVERY entity (worlds with a movable landmark, core.py:158-169)."""
    A = env.n
    W = len(seeds)
    E = len(env.world.entities)
    NV = E if all_vel else A
    arng = arng or np.random.RandomState(1234)
    dims = [env.observation_space[i].shape[0] for i in range(A)]
    out = {"seeds": np.array(seeds), "pos0": np.zeros((W, E, 2)), "vel0": np.zeros((W, NV, 2)),
           "act": np.zeros((T, W, A, 5)), "rew": np.zeros((T, W, A)), "done": np.zeros((T, W, A), bool),
           "pos": np.zeros((T, W, E, 2)), "vel": np.zeros((T, W, NV, 2))}
    if ids:
        out["ids"] = np.zeros((T, W, A), np.int64)
    for i in range(A):
        out["obs%d" % i] = np.zeros((T, W, dims[i]))
        out["obs_reset%d" % i] = np.zeros((W, dims[i]))
    info_keys = None
    for w, seed in enumerate(seeds):
        np.random.seed(int(seed))
        obs = env.reset()
        if squeeze_every and w % squeeze_every == squeeze_every - 1:
            for ent in env.world.entities:
                ent.state.p_pos = ent.state.p_pos * squeeze
            obs = [env._get_obs(a) for a in env.agents]
        out["pos0"][w], out["vel0"][w] = state_of(env, all_vel)
        for i in range(A):
            out["obs_reset%d" % i][w] = obs[i]
        for t in range(T):
            if ids:
                k = arng.randint(0, 5, size=A)
                out["ids"][t, w] = k
                act = [int(x) for x in k]
            else:
                act = []
                for i in range(A):
                    if soft_every and (t + w + i) % soft_every == soft_every - 1:
                        a = arng.uniform(-1, 1, 5)          # real-valued ("soft") action row
                    else:
                        a = np.eye(5)[arng.randint(0, 5)]
                    act.append(a)
                out["act"][t, w] = np.array(act)
                act = [a.copy() for a in act]
            obs, rew, done, info = env.step(act)
            for i in range(A):
                out["obs%d" % i][t, w] = obs[i]
            out["rew"][t, w] = np.array(rew, dtype=np.float64)
            out["done"][t, w] = done
            out["pos"][t, w], out["vel"][t, w] = state_of(env, all_vel)
            inf = info["n"]
            if name.startswith("simple_spread") and inf and inf[0] != {}:
                if info_keys is None:
                    info_keys = True
                    out["info_rew"] = np.zeros((T, W, A))
                    out["info_collisions"] = np.zeros((T, W, A), np.int32)
                    out["info_min_dists"] = np.zeros((T, W, A))
                    out["info_occupied"] = np.zeros((T, W, A), np.int32)
                for i in range(A):
                    r, c, md, oc = inf[i]
                    out["info_rew"][t, w, i] = r
                    out["info_collisions"][t, w, i] = c
                    out["info_min_dists"][t, w, i] = md
                    out["info_occupied"][t, w, i] = oc
            elif name == "simple_tag" and inf and inf[0] != {}:
                if info_keys is None:
                    info_keys = True
                    out["info_collisions"] = np.zeros((T, W, A), np.int32)
                out["info_collisions"][t, w] = np.array(inf, dtype=np.int32)
    return out


def main():
    t0 = time.time()
    only = sys.argv[1:]          # e.g. `gen_golden.py simple_spread_n64`: rewrite just that file
    if only == ["simple_spread_n64"]:
        jobs = [("simple_spread_n64", record_n64())]
        return write(jobs, t0)
    if only == ["simple_spread_n64_w64"]:
        return write([("simple_spread_n64_w64", record_n64_wide())], t0, compressed=True)
    jobs = []
    # C1: simple, 100 random-action steps (BASELINE.json configs[0]) -- plumbing check
    jobs.append(("simple", record("simple", make_env("simple"), list(range(16)), 100)))
    # simple_spread N=3 (configs[1], [4]); every 3rd world squeezed so that agents overlap
    jobs.append(("simple_spread", record("simple_spread", make_env("simple_spread", benchmark=True),
                                         list(range(48)), 25, squeeze_every=3)))
    # simple_tag (configs[2]); squeezed worlds put predators, prey and obstacles in contact
    jobs.append(("simple_tag", record("simple_tag", make_env("simple_tag", benchmark=True),
                                      list(range(48)), 25, squeeze_every=3, squeeze=0.25)))
    # simple_spread with integer action ids (environment.py:161-167; discrete_action_input)
    env = make_env("simple_spread", benchmark=True)
    env.discrete_action_input = True
    jobs.append(("simple_spread_ids", record("simple_spread_ids", env, list(range(100, 116)), 10,
                                             squeeze_every=2, ids=True)))
    jobs.append(("simple_spread_n64", record_n64()))
    # a mid-size N to pin the N-generic code (N=5)
    jobs.append(("simple_spread_n5", record("simple_spread_n5", spread_n(5), list(range(200, 212)), 12,
                                            squeeze_every=2, squeeze=0.4)))
    write(jobs, t0)


def record_n64():
    """simple_spread N=64 (configs[3]); 0.9 s per reference step: 8 worlds x 3 steps, every other world squeezed so
    that dozens of agents overlap (the contact path at N=64, not just the far field)."""
    return record("simple_spread_n64", spread_n(64), [7, 8, 9, 10, 11, 12, 13, 14], 3, squeeze_every=2, squeeze=0.5)


N64_WIDE_AGENTS = (0, 17, 63)


def record_n64_wide():
    """simple_spread N=64, 64 worlds x 3 steps (round 4: reference data for the REAL grid of the two-waves-per-world kernel --
    the GPU test places these worlds among 4096).  Full state, rewards and benchmark counts of every agent; observation rows
    (384 floats each) of agents 0, 17 and 63 only: every row of all 64 agents would be 38 MB."""
    data = record("simple_spread_n64", spread_n(64), list(range(300, 364)), 3, squeeze_every=2, squeeze=0.5)
    for k in list(data):
        if k.startswith("obs") and int(k.replace("obs_reset", "").replace("obs", "")) not in N64_WIDE_AGENTS:
            del data[k]
    data["obs_agents"] = np.array(N64_WIDE_AGENTS)
    return data


def write(jobs, t0, compressed=False):
    for name, data in jobs:
        path = os.path.join(HERE, name + ".npz")
        if compressed:
            np.savez_compressed(path, **data)
            print("%-24s %6.2f MB (compressed)" % (name, os.path.getsize(path) / 1e6))
            continue
        np.savez_compressed(path, **data)
        print("%-22s %8.1f KiB" % (name, os.path.getsize(path) / 1024.0))
    print("done in %.1f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
