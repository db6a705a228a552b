#!/usr/bin/env python3
"""Golden vectors for the six scenarios outside BASELINE.json's configs, recorded from
the UNMODIFIED reference (runs only in the build container; see gen_golden.py for the import recipe).

    python tests/golden/gen_golden_scenarios.py       # rewrites tests/golden/f3_<scenario>.npz

Per scenario (W worlds, T steps):
  seeds [W]            np.random.seed(seed) immediately before env.reset()
  choice [W,k]         the np.random.choice draws of reset_world as landmark indices (goal, key, ...)
  pos0/vel0            state after the reset (some worlds squeezed towards the origin: contacts)
  obs_reset{i} [W,D_i] observations of that state
  act{i} [T,W,d_i]     action rows fed to env.step: one-hot or real-valued; d_i = 5 (move), dim_c (speak)
                       or 5 + dim_c (MultiDiscrete: move and speak, environment.py:148-155)
  obs{i}, rew [T,W,A], pos, vel, c{i} [T,W,dim_c]   outputs / state after each step
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
from make_env import make_env  # noqa: E402


def choices_of(name, world):
    lm = world.landmarks
    if name in ("simple_adversary", "simple_push"):
        return [lm.index(world.agents[0].goal_a)]
    if name == "simple_speaker_listener":
        return [lm.index(world.agents[0].goal_b)]
    if name == "simple_reference":
        return [lm.index(world.agents[0].goal_b), lm.index(world.agents[1].goal_b)]
    if name == "simple_crypto":
        return [lm.index(world.agents[0].goal_a), int(np.argmax(world.agents[2].key))]
    return []


def action_dims(env):
    dims = []
    for agent in env.agents:
        d = 0
        if agent.movable:
            d += 5
        if not agent.silent:
            d += env.world.dim_c
        dims.append(d)
    return dims


def draw_action(agent, world, rng, soft):
    parts = []
    if agent.movable:
        parts.append(rng.uniform(-1, 1, 5) if soft else np.eye(5)[rng.randint(0, 5)])
    if not agent.silent:
        parts.append(rng.uniform(0, 1, world.dim_c) if soft else np.eye(world.dim_c)[rng.randint(0, world.dim_c)])
    return np.concatenate(parts)


def stage_world(name, w, world, rng):
    """Worlds staged after their reset so that the scenario's rare discrete branches are well populated (the reference
    still computes every output from the staged state): returns True when world w was touched.
    simple_push: every 2nd world has its two agents in contact.  simple_world_comm, by w % 8 (1 = squeezed by 0.3):
      2 prey in the boundary band (|x| in [0.88, 1.15])   3 prey within reach of an adversary   4 prey on a food item
      5 everybody fast (speed limits)                      6 agents gathered around the forests   7 = 3 and 5 together"""
    ag = world.agents
    if name == "simple_push" and w % 2 == 1:   # (instead of the squeeze: a contact lasts a step or two, so half the worlds start in one)
        th = rng.uniform(0, 2 * np.pi)
        ag[1].state.p_pos = ag[0].state.p_pos + rng.uniform(0.05, 0.11) * np.array([np.cos(th), np.sin(th)])
        return True
    if name != "simple_world_comm":
        return False
    k = w % 8
    disk = lambda r: (lambda th, rr: rr * np.array([np.cos(th), np.sin(th)]))(rng.uniform(0, 2 * np.pi), r * np.sqrt(rng.uniform(0, 1)))
    good = [a for a in ag if not a.adversary]
    advs = [a for a in ag if a.adversary]
    if k == 2:
        for a in good:
            p = a.state.p_pos.copy()
            p[rng.randint(0, 2)] = rng.choice([-1, 1]) * rng.uniform(0.88, 1.15)
            a.state.p_pos = p
    elif k == 4:
        for j, a in enumerate(good):
            a.state.p_pos = world.food[j % 2].state.p_pos + disk(0.07)
    elif k in (3, 5, 7):
        if k != 5:
            for a in good:
                a.state.p_pos = advs[rng.randint(0, len(advs))].state.p_pos + disk(0.13)
        if k != 3:
            for a in ag:
                a.state.p_vel = rng.uniform(-1.5, 1.5, 2)
    elif k == 6:
        for a in ag:
            a.state.p_pos = world.forests[rng.randint(0, 2)].state.p_pos + disk(0.45)
    else:
        return False
    return True


def record(name, seeds, T, squeeze_every=0, squeeze=0.3, customise=None, stage=False, env_factory=None):
    bench = name in ("simple_adversary", "simple_world_comm")   # benchmark_data as the info callback (make_env.py:36-43)
    env = env_factory(bench) if env_factory else make_env(name, benchmark=bench)   # (gen_golden_shapes.py: other team sizes)
    world = env.world
    consts = customise(env) if customise else {}     # gen_golden_custom.py: entity / world constants changed after make_world
    A, W, E = env.n, len(seeds), len(world.entities)
    rng = np.random.RandomState(4321)
    srng = np.random.RandomState(8765)      # staging draws: their own stream
    dims = [env.observation_space[i].shape[0] for i in range(A)]
    adims = action_dims(env)
    out = {"seeds": np.array(seeds), "pos0": np.zeros((W, E, 2)), "vel0": np.zeros((W, A, 2)), "staged": np.zeros(W, bool),
           "rew": np.zeros((T, W, A)), "pos": np.zeros((T, W, E, 2)), "vel": np.zeros((T, W, A, 2))}
    nch = None
    for i in range(A):
        out["obs%d" % i] = np.zeros((T, W, dims[i]))
        out["obs_reset%d" % i] = np.zeros((W, dims[i]))
        out["act%d" % i] = np.zeros((T, W, adims[i]))
        out["c%d" % i] = np.zeros((T, W, world.dim_c))
    for w, seed in enumerate(seeds):
        np.random.seed(int(seed))
        obs = env.reset()
        ch = choices_of(name, world)
        if nch is None:
            nch = len(ch)
            out["choice"] = np.zeros((W, nch), np.int64)
        out["choice"][w] = ch
        if squeeze_every and w % squeeze_every == squeeze_every - 1:
            for ent in world.entities:
                ent.state.p_pos = ent.state.p_pos * squeeze
            out["staged"][w] = True
        if stage and stage_world(name, w, world, srng):
            out["staged"][w] = True
        if out["staged"][w]:
            obs = [env._get_obs(a) for a in env.agents]
        out["pos0"][w] = np.array([e.state.p_pos for e in world.entities])
        out["vel0"][w] = np.array([a.state.p_vel for a in world.agents])
        for i in range(A):
            out["obs_reset%d" % i][w] = obs[i]
        for t in range(T):
            act = []
            for i, agent in enumerate(env.agents):
                a = draw_action(agent, world, rng, soft=((t + w + i) % 4 == 3))
                out["act%d" % i][t, w] = a
                act.append(a.copy())
            obs, rew, done, info = env.step(act)
            if bench:    # simple_adversary.py:57-67 (adversary: a squared distance; good agents: L + 1 of them), simple_world_comm.py:115-124
                inf = info["n"]
                if name == "simple_adversary":
                    if "info_adv" not in out:
                        nadv = sum(1 for a in env.agents if a.adversary)
                        out["info_adv"] = np.zeros((T, W, nadv))
                        out["info_good"] = np.zeros((T, W, A - nadv, len(world.landmarks) + 1))
                    nadv = sum(1 for a in env.agents if a.adversary)
                    out["info_adv"][t, w, :] = inf[:nadv]
                    out["info_good"][t, w] = np.array([list(x) for x in inf[nadv:]])
                else:
                    if "info_collisions" not in out:
                        out["info_collisions"] = np.zeros((T, W, A), np.int32)
                    out["info_collisions"][t, w] = np.array(inf, dtype=np.int32)
            for i in range(A):
                out["obs%d" % i][t, w] = obs[i]
                out["c%d" % i][t, w] = world.agents[i].state.c
            out["rew"][t, w] = np.array(rew, dtype=np.float64)
            out["pos"][t, w] = np.array([e.state.p_pos for e in world.entities])
            out["vel"][t, w] = np.array([a.state.p_vel for a in world.agents])
    out.update(consts)
    return out


def coverage(name, d):
    """Share of the recorded (world, step[, agent / pair]) samples that take each discrete branch of the scenario's
    callbacks -- printed here and asserted by tests/test_oracle_golden.py (every branch >= 5 %)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import spec as ospec
    from oracle.mpe_f3 import branch_coverage
    return branch_coverage(ospec.by_name(name), d)


def main():
    t0 = time.time()
    # W worlds x T steps per scenario; every `sq`-th world squeezed towards the origin after its reset (contacts, forests)
    jobs = [("simple_adversary", 256, 8, 0), ("simple_push", 384, 4, 0), ("simple_speaker_listener", 256, 8, 0),
            ("simple_reference", 256, 8, 0), ("simple_crypto", 256, 8, 0), ("simple_world_comm", 384, 4, 8)]
    for name, W, T, sq in jobs:
        data = record(name, list(range(300, 300 + W)), T, squeeze_every=sq, stage=True)
        path = os.path.join(HERE, "f3_" + name + ".npz")
        np.savez_compressed(path, **data)
        print("%-26s %8.1f KiB  choices %s" % (name, os.path.getsize(path) / 1024.0, data["choice"][:6].tolist()))
        print("    coverage: " + ", ".join("%s %.1f%%" % (k, 100 * v) for k, v in coverage(name, data).items()))
    print("done in %.1f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
