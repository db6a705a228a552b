"""A reference-STYLE scenario file (test fixture, written for this repo -- not one of the reference's nine): the contract
of multiagent/scenario.py:4-10 and the README's "Creating new environments" -- `from multiagent.core import ...`,
`make_world(self)`, `reset_world(self, world)`, NumPy per-world `reward` / `observation`.

survey: three drones map three sites.  Written the way scenario authors write when they are NOT thinking of a tracer: the
observation is filled into a preallocated `np.zeros` row slice by slice, positions are quantised to a grid (`np.floor`), wrapped
(`%`), offsets are compressed (`np.sign(d) * np.abs(d) ** 1.5`), flags are built with `float(test)` / `int(test)`, distances
with `np.hypot` / `math.hypot`, rewards rounded with `np.round`.  The rally point of a world is drawn in reset_world and kept as plain
coordinates on the world object (`world.rally`) -- not an entity, not in anybody's state vector -- and read by both callbacks.
Everybody is silent; the reward is shared.
"""
import math

import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario

GRID = 4.0


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.dim_c = 2
        world.collaborative = True
        world.agents = [Agent() for _ in range(3)]
        for i, agent in enumerate(world.agents):
            agent.name = "drone %d" % i
            agent.collide = True
            agent.silent = True
            agent.size = 0.08
        world.landmarks = [Landmark() for _ in range(3)]
        for i, lm in enumerate(world.landmarks):
            lm.name = "site %d" % i
            lm.collide = False
            lm.movable = False
            lm.size = 0.05
        self.reset_world(world)
        return world

    def reset_world(self, world):
        world.rally = np.random.uniform(-0.5, +0.5, world.dim_p)
        for agent in world.agents:
            agent.color = np.array([0.35, 0.35, 0.85])
            agent.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.color = np.array([0.25, 0.25, 0.25])
            lm.state.p_pos = np.random.uniform(-0.8, +0.8, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)

    def touching(self, a, b):
        return math.hypot(a.state.p_pos[0] - b.state.p_pos[0], a.state.p_pos[1] - b.state.p_pos[1]) < a.size + b.size

    def reward(self, agent, world):
        rew = 0.0
        for lm in world.landmarks:
            dists = [np.hypot(a.state.p_pos[0] - lm.state.p_pos[0], a.state.p_pos[1] - lm.state.p_pos[1]) for a in world.agents]
            nearest = min(dists)
            rew -= np.round(nearest, 2)                 # centimetres
            rew += 0.25 * int(nearest < 0.1)            # a site is mapped while a drone hovers over it
        for a in world.agents:
            for b in world.agents:
                if a is not b:
                    rew -= 0.5 * float(self.touching(a, b))
        rew -= 0.1 * np.linalg.norm(agent.state.p_pos - world.rally)
        return rew

    def observation(self, agent, world):
        n = len(world.landmarks)
        row = np.zeros(2 + 2 + 2 + 2 * n + 2 + 2)
        row[0:2] = agent.state.p_vel
        row[2:4] = np.floor(agent.state.p_pos * GRID) / GRID           # the grid cell the drone is over
        row[4:6] = (agent.state.p_pos + 1.0) % (1.0 / GRID) * GRID      # where in that cell, 0 .. 1
        for k, lm in enumerate(world.landmarks):
            d = lm.state.p_pos - agent.state.p_pos
            row[6 + 2 * k:8 + 2 * k] = np.sign(d) * np.abs(d) ** 1.5    # compressed offsets: fine near, coarse far
        # (thresholds on POSITIONS: velocities under one-hot moves sit exactly on round numbers -- 0.5 after one step from rest --
        #  where an fp32 and an fp64 evaluation may legitimately land on different sides)
        home = np.hypot(agent.state.p_pos[0], agent.state.p_pos[1])
        row[-4] = float(home > 0.5)
        row[-3] = math.floor(home * 10.0) / 10.0
        row[-2:] = world.rally - agent.state.p_pos
        return row
