"""simple_reference: two agents each know the OTHER's goal landmark and have to talk it over
(reference: multiagent/scenarios/simple_reference.py).  Agents move and speak: MultiDiscrete actions.
Generic path."""
import torch

from .. import _abi
from ..core import World, Agent, Landmark, EntityChoice
from ..scenario import BaseScenario
from . import _util as U


class Scenario(BaseScenario):
    kind = _abi.MPE_SCN_REFERENCE   # fused kernel (wave-per-agent family)
    landmark_range = 1.0

    def make_world(self, batch_size=1, device=None):
        world = World(batch_size, device)          # simple_reference.py:6-25
        world.dim_c = 10
        world.choice_pops = [3, 3]                 # agents[0].goal_b, agents[1].goal_b (:35, :37)
        world.collaborative = True
        world.agents = [Agent() for _ in range(2)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
        world.landmarks = [Landmark() for _ in range(3)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        world.allocate()
        self._world = world
        self._apply(world)
        return world

    @property
    def goal_index(self):                          # [B, 2] long
        return self._world.choice_i32.t().long()

    def reset_world(self, world, mask=None, seeds=None):   # simple_reference.py:27-55: two choices, then positions
        n = len(world.landmarks)
        idx = world.reset_uniform(self.landmark_range, mask, choices=[n, n], seeds=seeds)
        m = None if mask is None else torch.as_tensor(mask, device=world.device).bool()[:, None]
        self.set_goal(world, World.merge_choice(self.goal_index, idx, m))

    def set_goal(self, world, index):
        """index [B, 2]: agents[0].goal_b, agents[1].goal_b."""
        world.choice_i32.copy_(torch.as_tensor(index, device=world.device).reshape(world.batch_size, 2).t().int())
        self._apply(world)

    def _apply(self, world):
        world.landmarks[0].color = U.const(world, [0.75, 0.25, 0.25])
        world.landmarks[1].color = U.const(world, [0.25, 0.75, 0.25])
        world.landmarks[2].color = U.const(world, [0.25, 0.25, 0.75])
        a0, a1 = world.agents
        a0.goal_a, a0.goal_b = a1, EntityChoice(world, world.landmarks, world.choice_i32[0])
        a1.goal_a, a1.goal_b = a0, EntityChoice(world, world.landmarks, world.choice_i32[1])
        U.assign(a1, "color", a0.goal_b.color)   # :43-44 (goal_a.color = goal_b.color; rendering only)
        U.assign(a0, "color", a1.goal_b.color)

    def reward(self, agent, world):                # simple_reference.py:57-61
        if agent.goal_a is None or agent.goal_b is None:
            return U.zeros(world)
        return -U.dist2(agent.goal_a, agent.goal_b)

    def observation(self, agent, world):           # simple_reference.py:63-83
        goal_color = [U.zeros(world, world.dim_color), U.zeros(world, world.dim_color)]
        if agent.goal_b is not None:
            goal_color[1] = agent.goal_b.color
        entity_pos = [entity.state.p_pos - agent.state.p_pos for entity in world.landmarks]
        comm = [other.state.c for other in world.agents if other is not agent]
        return torch.cat([agent.state.p_vel] + entity_pos + [goal_color[1]] + comm, dim=1)
