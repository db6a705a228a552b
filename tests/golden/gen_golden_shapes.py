#!/usr/bin/env python3
"""Golden vectors for simple_adversary / simple_world_comm at team sizes OTHER than the reference's make_world, recorded
from the reference's own callbacks (build container only; import recipe in gen_golden.py).

    python tests/golden/gen_golden_shapes.py       # rewrites tests/golden/shape_<scenario>_<A>_<n_adv>.npz

The reference's make_world fixes the counts in local variables (simple_adversary.py:9-13: 3 agents, 1 adversary;
simple_world_comm.py:10-16: 4 + 2), but reset_world / reward / observation / benchmark_data loop over world.agents,
good_agents() and adversaries() and are written for any team sizes.  So the world is built by the UNMODIFIED
make_world, its agent (and, for simple_adversary, landmark) lists are then shortened or extended with deep copies of
their own members, and the unmodified callbacks run on that world through the reference's MultiAgentEnv -- the same
construction make_env.py:33-44 performs.  Layout of the files: gen_golden_scenarios.py.
"""
import copy
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_scenarios as G  # noqa: E402  (sets up the stub gym + reference import path)
import numpy as np  # noqa: E402

import multiagent.scenarios as scenarios  # noqa: E402
from multiagent.environment import MultiAgentEnv  # noqa: E402


def resized(lst, n, template):
    """The first n members of lst, topped up with deep copies of lst[template]."""
    out = list(lst[:n])
    while len(out) < n:
        out.append(copy.deepcopy(lst[template]))
    return out


def factory(name, n_agents, n_adv):
    def make(bench):
        sc = scenarios.load(name + ".py").Scenario()
        world = sc.make_world()
        advs = [a for a in world.agents if a.adversary]
        good = [a for a in world.agents if not a.adversary]
        # (a copied adversary must be a follower: simple_world_comm's agent 0 is the only leader / speaker)
        world.agents = resized(advs, n_adv, len(advs) - 1) + resized(good, n_agents - n_adv, 0)
        for i, a in enumerate(world.agents):
            a.name = 'agent %d' % i
        if name == "simple_adversary":          # num_landmarks = num_agents - 1 (:13); reset_world reads world.num_agents (:38)
            world.num_agents = n_agents
            world.landmarks = resized(world.landmarks, n_agents - 1, 0)
            for i, l in enumerate(world.landmarks):
                l.name = 'landmark %d' % i
        sc.reset_world(world)
        return MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation, sc.benchmark_data if bench else None)
    return make


sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.spec import TEAM_SIZE_VARIANTS as SHAPES, team_size_spec  # noqa: E402  (the list the tests iterate over)


def main():
    from oracle.mpe_f3 import branch_coverage
    for name, A, nadv in SHAPES:
        wc = name == "simple_world_comm"
        W, T = (64, 3) if wc else (40, 3)
        data = G.record(name, list(range(700, 700 + W)), T, squeeze_every=8 if wc else 0, stage=True,
                        env_factory=factory(name, A, nadv))
        data["n_agents"], data["n_adversaries"] = np.int64(A), np.int64(nadv)
        path = os.path.join(HERE, "shape_%s_%d_%d.npz" % (name, A, nadv))
        np.savez_compressed(path, **data)
        cov = branch_coverage(team_size_spec(name, A, nadv), data)
        print("%-40s %7.1f KiB  coverage: %s" % (os.path.basename(path), os.path.getsize(path) / 1024.0,
                                                ", ".join("%s %.1f%%" % (k, 100 * v) for k, v in cov.items())))


if __name__ == "__main__":
    main()
