"""A reference-STYLE scenario file (test fixture, written for this repo -- not one of the reference's nine): the contract
of multiagent/scenario.py:4-10 and the README's "Creating new environments" -- `from multiagent.core import ...`,
`make_world(self)`, `reset_world(self, world)`, NumPy per-world `reward` / `observation`.

scatter: three agents each claim the nearest of three (solid) beacons; agent 0, the courier, must reach the beacon that was picked
for this world.  Its reset_world is NOT a formula of its random draws: entities are placed by REJECTION sampling (`while` the spot
is taken: draw again -- a data-dependent number of draws), and the agents start with a small normal-distributed velocity.  Such a
reset cannot be traced into a program; it stays what it is -- Python, run per world at reset time, its np.random.choice followed as
a per-world pick -- while observation and reward are traced into the step kernel as for any other file.
"""
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario

CLEARANCE = 0.3


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.agents = [Agent() for _ in range(3)]
        for i, agent in enumerate(world.agents):
            agent.name = "agent %d" % i
            agent.courier = i == 0
            agent.collide = True
            agent.silent = True
            agent.size = 0.08
        world.landmarks = [Landmark() for _ in range(3)]
        for i, beacon in enumerate(world.landmarks):
            beacon.name = "beacon %d" % i
            beacon.collide = True
            beacon.movable = False
            beacon.size = 0.12
        self.reset_world(world)
        return world

    def reset_world(self, world):
        world.goal = np.random.choice(world.landmarks)
        placed = []
        for entity in world.landmarks + world.agents:
            entity.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            while any(np.linalg.norm(entity.state.p_pos - other.state.p_pos) < CLEARANCE for other in placed):
                entity.state.p_pos = np.random.uniform(-1, +1, world.dim_p)          # taken: draw again
            entity.state.p_vel = np.zeros(world.dim_p)
            placed.append(entity)
        for agent in world.agents:
            agent.state.p_vel = np.random.randn(world.dim_p) * 0.05
            agent.state.c = np.zeros(world.dim_c)

    def reward(self, agent, world):
        dists = [np.linalg.norm(agent.state.p_pos - beacon.state.p_pos) for beacon in world.landmarks]
        rew = -np.linalg.norm(agent.state.p_pos - world.goal.state.p_pos) if agent.courier else -min(dists)
        for other in world.agents:
            if other is not agent and np.linalg.norm(other.state.p_pos - agent.state.p_pos) < other.size + agent.size:
                rew -= 1.0
        return rew

    def observation(self, agent, world):
        beacons = [beacon.state.p_pos - agent.state.p_pos for beacon in world.landmarks]
        others = [other.state.p_pos - agent.state.p_pos for other in world.agents if other is not agent]
        return np.concatenate([agent.state.p_vel, agent.state.p_pos] + beacons + others + [world.goal.state.p_pos - agent.state.p_pos])
