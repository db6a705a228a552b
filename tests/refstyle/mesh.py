"""A reference-STYLE scenario file (test fixture, written for this repo -- not one of the reference's nine): the contract
of multiagent/scenario.py:4-10 and the README's "Creating new environments" -- `from multiagent.core import ...`,
`make_world(self)`, `reset_world(self, world)`, NumPy per-world `reward` / `observation`.

mesh: four agents cover four landmarks, written by somebody who thinks in arrays -- a distance matrix, comparisons of whole
arrays (`D < 0.2`), reductions along an axis as methods (`D.min(axis=0)`, `.all()`), `continue` in loops, nested early returns,
`a < x < b`.  NumPy's own `<` on an array of symbolic values would ask every element for its truth (a fork per element: 2^16 paths
for `D < 0.2` alone); the tracer's predicated twin of this file has one path per callback.
"""
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    """Written by somebody who thinks in arrays: distance matrices, comparisons of whole arrays, reductions along an axis."""
    def make_world(self):
        world = World()
        world.agents = [Agent() for _ in range(4)]
        for i, a in enumerate(world.agents):
            a.name, a.silent, a.size = "agent %d" % i, True, 0.08
        world.landmarks = [Landmark() for _ in range(4)]
        for l in world.landmarks:
            l.movable, l.collide, l.size = False, False, 0.1
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for e in world.agents + world.landmarks:
            e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            e.state.p_vel = np.zeros(world.dim_p)
        for a in world.agents:
            a.state.c = np.zeros(world.dim_c)

    def zone(self, p):                      # nested early returns
        if abs(p[0]) < 0.5:
            if abs(p[1]) < 0.5:
                return 2.0
            return 1.0
        elif abs(p[0]) < 0.8:
            return 0.5
        else:
            return 0.0

    def reward(self, agent, world):
        X = np.array([a.state.p_pos for a in world.agents])
        Y = np.array([l.state.p_pos for l in world.landmarks])
        D = np.sqrt(((X[:, None, :] - Y[None, :, :]) ** 2).sum(-1))          # [agents, landmarks]
        rew = -D.min(axis=0).sum()                                           # the nearest agent of every landmark
        rew += 0.1 * (D < 0.2).sum() + 0.05 * np.count_nonzero(D.min(axis=1) < 0.1)
        if (D.min(axis=0) < 0.15).all():
            rew += 5.0
        if np.any(np.abs(X) > 0.95):
            rew -= 1.0
        for a in world.agents:
            if a is agent:
                continue
            gap = np.linalg.norm(a.state.p_pos - agent.state.p_pos)
            if gap > 0.5:
                continue
            rew -= 0.5 - gap
        if -0.25 < agent.state.p_pos[0] < 0.25:
            rew += 0.01
        return rew + self.zone(agent.state.p_pos) * 0.01

    def observation(self, agent, world):
        Y = np.array([l.state.p_pos for l in world.landmarks]) - agent.state.p_pos
        near = np.linalg.norm(Y, axis=1) < 0.6
        row = np.concatenate([agent.state.p_vel, agent.state.p_pos, (Y * near[:, None]).reshape(-1), Y.max(axis=0), np.clip(Y, -0.5, 0.5).min(axis=0)])
        return row.astype(np.float32)          # (the gym habit: observations in single precision)
