"""A reference-STYLE scenario file (test fixture, written for this repo): communication.

relay: a fixed beacon (agent 0, cannot move, speaks dim_c = 4 words) knows which of two landmarks is the target; a
runner (agent 1, moves AND speaks) must reach it.  Shared reward (world.collaborative).  Action rows: beacon = a word,
runner = MultiDiscrete move + word (environment.py:148-155).
"""
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.dim_c = 4
        world.collaborative = True
        beacon, runner = Agent(), Agent()
        beacon.name, runner.name = "beacon", "runner"
        beacon.movable, beacon.silent, beacon.collide, beacon.size = False, False, True, 0.2
        runner.movable, runner.silent, runner.collide, runner.size = True, False, True, 0.07
        runner.accel = 3.5
        world.agents = [beacon, runner]
        world.landmarks = [Landmark() for _ in range(2)]
        for i, lm in enumerate(world.landmarks):
            lm.name = "target %d" % i
            lm.collide = False
            lm.movable = False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        world.target = int(np.random.randint(0, 2))
        world.agents[0].color = np.array([0.25, 0.25, 0.25])
        world.agents[1].color = np.array([0.1, 0.1, 0.1])
        world.agents[1].color[world.target] += 0.8
        for agent in world.agents:
            agent.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.state.p_pos = np.random.uniform(-0.9, +0.9, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)

    def reward(self, agent, world):
        runner = world.agents[1]
        d2 = np.sum(np.square(runner.state.p_pos - world.landmarks[world.target].state.p_pos))
        chatter = 0.01 * np.sum(agent.state.c)
        return -d2 - chatter

    def observation(self, agent, world):
        beacon, runner = world.agents
        if agent is beacon:
            onehot = np.zeros(2)
            onehot[world.target] = 1.0
            return np.concatenate([onehot, runner.state.p_pos - beacon.state.p_pos, runner.state.c])
        rel = [lm.state.p_pos - runner.state.p_pos for lm in world.landmarks]
        return np.concatenate([runner.state.p_vel] + rel + [beacon.state.p_pos - runner.state.p_pos, beacon.state.c])
