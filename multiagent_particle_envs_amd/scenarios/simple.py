"""simple: one agent, one landmark, reward = -|agent - landmark|^2
(reference: multiagent/scenarios/simple.py).  Fused kernel kind MPE_SCN_SIMPLE."""
import torch

from .. import _abi
from ..core import World, Agent, Landmark
from ..scenario import BaseScenario


class Scenario(BaseScenario):
    kind = _abi.MPE_SCN_SIMPLE
    landmark_range = 1.0

    def make_world(self, batch_size=1, device=None):
        world = World(batch_size, device)          # simple.py:6-22
        world.agents = [Agent() for _ in range(1)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
            agent.silent = True
        world.landmarks = [Landmark() for _ in range(1)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        world.allocate()
        return world

    def reset_world(self, world, mask=None, seeds=None):       # simple.py:24-39
        world.reset_uniform(self.landmark_range, mask, seeds=seeds)

    # per-agent callbacks (generic path; the env uses the fused kernel instead when unmodified)
    def reward(self, agent, world):                # simple.py:41-43
        d = agent.state.p_pos - world.landmarks[0].state.p_pos
        return -(d * d).sum(dim=1)

    def observation(self, agent, world):           # simple.py:45-50
        entity_pos = [lm.state.p_pos - agent.state.p_pos for lm in world.landmarks]
        return torch.cat([agent.state.p_vel] + entity_pos, dim=1)
