"""A reference-STYLE scenario file (test fixture, written for this repo -- not one of the reference's nine): the contract
of multiagent/scenario.py:4-10 and the README's "Creating new environments" -- `from multiagent.core import ...`,
`make_world(self)`, `reset_world(self, world)`, NumPy per-world `reward` / `observation` / `benchmark_data`.

herd: a shepherd (agent 0, big, slow) pushes two sheep (small, fast, speed-limited) towards the pen -- one of three
landmarks, picked per world at reset with np.random.choice.  Physical only (everybody silent); all agents collide.
"""
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.dim_c = 2
        world.damping = 0.3
        world.contact_force = 150.0
        world.agents = [Agent() for _ in range(3)]
        for i, agent in enumerate(world.agents):
            agent.name = "shepherd" if i == 0 else "sheep %d" % i
            agent.shepherd = i == 0
            agent.collide = True
            agent.silent = True
            agent.size = 0.12 if i == 0 else 0.06
            agent.accel = 2.5 if i == 0 else 4.5
            agent.max_speed = None if i == 0 else 1.1
            agent.initial_mass = 2.0 if i == 0 else 0.8
        world.landmarks = [Landmark() for _ in range(3)]
        for i, lm in enumerate(world.landmarks):
            lm.name = "pen %d" % i
            lm.collide = False
            lm.movable = False
            lm.size = 0.1
        self.reset_world(world)
        return world

    def reset_world(self, world):
        pen = np.random.choice(world.landmarks)
        for i, lm in enumerate(world.landmarks):
            lm.color = np.array([0.2, 0.2, 0.2])
            lm.color[i] += 0.6
        for agent in world.agents:
            agent.goal = pen
            agent.color = pen.color * (0.5 if agent.shepherd else 1.0)
            agent.state.p_pos = np.random.uniform(-0.8, +0.8, world.dim_p)
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)

    def touching(self, a, b):
        d = a.state.p_pos - b.state.p_pos
        return np.sqrt(np.sum(np.square(d))) < a.size + b.size

    def reward(self, agent, world):
        sheep = [a for a in world.agents if not a.shepherd]
        far = sum(np.sqrt(np.sum(np.square(s.state.p_pos - agent.goal.state.p_pos))) for s in sheep)
        if agent.shepherd:
            return -far
        rew = -np.sum(np.square(agent.state.p_pos - agent.goal.state.p_pos))
        if self.touching(agent, world.agents[0]):
            rew -= 2.0
        return rew

    def benchmark_data(self, agent, world):
        hits = sum(1 for a in world.agents if a is not agent and self.touching(a, agent))
        return (self.reward(agent, world), hits)

    def observation(self, agent, world):
        rel = [e.state.p_pos - agent.state.p_pos for e in world.landmarks]
        others = [o.state.p_pos - agent.state.p_pos for o in world.agents if o is not agent]
        vels = [o.state.p_vel for o in world.agents if o is not agent and not o.shepherd]
        head = [agent.state.p_vel, agent.state.p_pos]
        if not agent.shepherd:
            head = head + [agent.goal.color]
        return np.concatenate(head + rel + others + vels)
