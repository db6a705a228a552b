"""A reference-STYLE scenario file (test fixture, written for this repo -- not one of the reference's nine): the contract
of multiagent/scenario.py:4-10 and the README's "Creating new environments" -- `from multiagent.core import ...`,
`make_world(self)`, `reset_world(self, world)`, NumPy per-world `reward` / `observation`.

convoy: a scout (agent 0, fast) must reach the depot -- one of two depots, picked per world at reset -- while three trucks keep
formation behind it: each truck wants to stay between 0.25 and 0.6 of the scout, not to bump into rocks or into each other, and
not to leave the road (|y| < 0.8).  Everybody is silent.  reset_world places every agent uniformly on [-1, 1)^2 and every
landmark on [-0.9, 0.9)^2 -- the placement World.reset_uniform draws on the device.
"""
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.dim_c = 2
        world.collaborative = False
        world.agents = [Agent() for _ in range(4)]
        for i, agent in enumerate(world.agents):
            agent.name = "scout" if i == 0 else "truck %d" % i
            agent.scout = i == 0
            agent.collide = True
            agent.silent = True
            agent.size = 0.05 if agent.scout else 0.09
            agent.accel = 4.0 if agent.scout else 3.0
            agent.max_speed = 1.3 if agent.scout else 0.9
        world.landmarks = [Landmark() for _ in range(4)]
        for i, lm in enumerate(world.landmarks):
            lm.name = "depot %d" % i if i < 2 else "rock %d" % (i - 2)
            lm.depot = i < 2
            lm.collide = not lm.depot
            lm.movable = False
            lm.size = 0.06 if lm.depot else 0.15
        world.depots = world.landmarks[:2]
        world.rocks = world.landmarks[2:]
        self.reset_world(world)
        return world

    def reset_world(self, world):
        world.goal = np.random.choice(world.depots)
        for lm in world.landmarks:
            lm.color = np.array([0.3, 0.3, 0.3])
        world.goal.color = np.array([0.2, 0.8, 0.2])
        for agent in world.agents:
            agent.color = np.array([0.8, 0.6, 0.2]) if agent.scout else np.array([0.3, 0.3, 0.8])
            agent.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.state.p_pos = np.random.uniform(-0.9, +0.9, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)

    def dist(self, a, b):
        return np.sqrt(np.sum(np.square(a.state.p_pos - b.state.p_pos)))

    def bumped(self, a, b):
        return self.dist(a, b) < a.size + b.size

    def reward(self, agent, world):
        scout = world.agents[0]
        if agent.scout:
            rew = -self.dist(scout, world.goal)
            if self.bumped(scout, world.goal):
                rew += 5.0
            return rew
        gap = self.dist(agent, scout)
        rew = 0.0
        if gap < 0.25:
            rew -= (0.25 - gap) * 4
        elif gap > 0.6:
            rew -= min(gap - 0.6, 1.0)
        for rock in world.rocks:
            if self.bumped(agent, rock):
                rew -= 2.0
        for other in world.agents:
            if other is not agent and not other.scout and self.bumped(agent, other):
                rew -= 1.0
        off_road = abs(agent.state.p_pos[1]) - 0.8
        if off_road > 0:
            rew -= 10 * off_road
        return rew

    def observation(self, agent, world):
        rocks = [r.state.p_pos - agent.state.p_pos for r in world.rocks]
        others = [o.state.p_pos - agent.state.p_pos for o in world.agents if o is not agent]
        head = [agent.state.p_vel, agent.state.p_pos]
        if agent.scout:      # only the scout knows which depot it is
            head = head + [world.goal.state.p_pos - agent.state.p_pos] + [d.color for d in world.depots]
        else:
            head = head + [world.agents[0].state.p_vel]
        return np.concatenate(head + rocks + others)
