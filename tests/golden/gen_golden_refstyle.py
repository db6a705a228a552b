#!/usr/bin/env python3
"""Golden vectors for the reference-STYLE fixture scenarios (tests/refstyle/*.py), recorded by running those files through the
UNMODIFIED reference's own MultiAgentEnv (build container only; import recipe as in gen_golden.py).

    python tests/golden/gen_golden_refstyle.py        # rewrites tests/golden/refstyle_<name>.npz

The fixture files are written against the reference's plug-in contract (scenario.py:4-10); here the REFERENCE steps them
(its core.py physics, its environment.py), on the GPU box this package steps the same files (refstyle.RefScenarioAdapter
over mpe_world_step) and must produce the same numbers.  Per scenario (W worlds, T steps, free-running in the reference):
  seeds [W]                np.random.seed(seed) immediately before env.reset()
  pos0 / vel0 [W,E,2]      state after the reset (every 3rd world squeezed towards the origin: contacts)
  obs_reset{i} [W,D_i]     observations of that state
  act{i} [T,W,d_i]         policy agent i's action row (move 5 and / or word dim_c; one-hot, every 4th real-valued)
  obs{i} [T,W,D_i], rew [T,W,n], done [T,W,n], info{k} [T,W,n], pos / vel [T,W,E,2], c{i} [T,W,dim_c] (all agents)
"""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["SUPPRESS_MA_PROMPT"] = "1"
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "_gym_stub"), "/root/reference"]

import numpy as np  # noqa: E402
import warnings  # noqa: E402

warnings.filterwarnings("ignore")
from multiagent.environment import MultiAgentEnv  # noqa: E402  (the reference's env, environment.py:9)

FIXTURES = os.path.join(os.path.dirname(HERE), "refstyle")
NAMES = ("herd", "relay", "patrol", "convoy", "survey", "mesh", "scatter")


def load(name):
    spec = importlib.util.spec_from_file_location("refstyle_fixture_" + name, os.path.join(FIXTURES, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Scenario()


def make(name, benchmark=True, done=True):
    sc = load(name)
    world = sc.make_world()
    return MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation,
                         sc.benchmark_data if benchmark and hasattr(sc, "benchmark_data") else None,
                         sc.done if done and hasattr(sc, "done") else None)


def record(name, seeds, T):
    env = make(name)
    world = env.world
    n, W, E, A = env.n, len(seeds), len(world.entities), len(world.agents)
    rng = np.random.RandomState(2468)
    dims = [env.observation_space[i].shape[0] for i in range(n)]
    adims = [(5 if a.movable else 0) + (world.dim_c if not a.silent else 0) for a in env.agents]
    out = {"seeds": np.array(seeds), "pos0": np.zeros((W, E, 2)), "vel0": np.zeros((W, E, 2)),
           "rew": np.zeros((T, W, n)), "done": np.zeros((T, W, n), bool), "pos": np.zeros((T, W, E, 2)), "vel": np.zeros((T, W, E, 2))}
    for i in range(n):
        out["obs%d" % i] = np.zeros((T, W, dims[i]))
        out["obs_reset%d" % i] = np.zeros((W, dims[i]))
        out["act%d" % i] = np.zeros((T, W, adims[i]))
    for i in range(A):
        out["c%d" % i] = np.zeros((T, W, world.dim_c))
    ninfo = None
    for w, seed in enumerate(seeds):
        np.random.seed(int(seed))
        obs = env.reset()
        if w % 3 == 2:
            for ent in world.entities:
                ent.state.p_pos = ent.state.p_pos * 0.3
            obs = [env._get_obs(a) for a in env.agents]
        out["pos0"][w] = np.array([e.state.p_pos for e in world.entities])
        out["vel0"][w] = np.array([e.state.p_vel for e in world.entities])
        for i in range(n):
            out["obs_reset%d" % i][w] = obs[i]
        for t in range(T):
            acts = []
            for i, a in enumerate(env.agents):
                soft = (t + w + i) % 4 == 3
                parts = []
                if a.movable:
                    parts.append(rng.uniform(-1, 1, 5) if soft else np.eye(5)[rng.randint(0, 5)])
                if not a.silent:
                    parts.append(rng.uniform(0, 1, world.dim_c) if soft else np.eye(world.dim_c)[rng.randint(0, world.dim_c)])
                acts.append(np.concatenate(parts))
                out["act%d" % i][t, w] = acts[-1]
            obs, rew, done, info = env.step([a.copy() for a in acts])
            for i in range(n):
                out["obs%d" % i][t, w] = obs[i]
            out["rew"][t, w] = np.array(rew, np.float64)
            out["done"][t, w] = np.array(done, bool)
            out["pos"][t, w] = np.array([e.state.p_pos for e in world.entities])
            out["vel"][t, w] = np.array([e.state.p_vel for e in world.entities])
            for i, a in enumerate(world.agents):
                out["c%d" % i][t, w] = a.state.c
            vals = [v if isinstance(v, tuple) else (v,) for v in info["n"]]
            if vals and not isinstance(info["n"][0], dict):
                if ninfo is None:
                    ninfo = len(vals[0])
                    for k in range(ninfo):
                        out["info%d" % k] = np.zeros((T, W, n))
                for k in range(ninfo):
                    out["info%d" % k][t, w] = [v[k] for v in vals]
    return out


def main():
    only = sys.argv[1:]
    for name in (only or NAMES):
        data = record(name, list(range(500, 512)), 12)
        np.savez(os.path.join(HERE, "refstyle_%s.npz" % name), **data)
        print(name, {k: v.shape for k, v in data.items() if k in ("rew", "pos", "obs0", "info0")},
              "done rate %.2f" % data["done"].mean(), "max |vel| %.2f" % np.abs(data["vel"]).max())


if __name__ == "__main__":
    main()
