"""simple_tag: predators chase a faster prey around colliding obstacles
(reference: multiagent/scenarios/simple_tag.py).  Fused kernel kind MPE_SCN_TAG."""
import torch

from .. import _abi
from ..core import World, Agent, Landmark
from ..scenario import BaseScenario


class Scenario(BaseScenario):
    kind = _abi.MPE_SCN_TAG
    landmark_range = 0.9                           # simple_tag.py:53
    num_adversaries = 3

    def make_world(self, batch_size=1, device=None, num_good_agents=1, num_adversaries=3, num_landmarks=2):
        world = World(batch_size, device)          # simple_tag.py:7-36
        world.dim_c = 2
        self.num_adversaries = num_adversaries
        num_agents = num_adversaries + num_good_agents
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.adversary = True if i < num_adversaries else False
            agent.size = 0.075 if agent.adversary else 0.05
            agent.accel = 3.0 if agent.adversary else 4.0
            agent.max_speed = 1.0 if agent.adversary else 1.3
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = True
            landmark.movable = False
            landmark.size = 0.2
            landmark.boundary = False
        world.allocate()
        return world

    def reset_world(self, world, mask=None, seeds=None):       # simple_tag.py:39-54
        world.reset_uniform(self.landmark_range, mask, seeds=seeds)

    def good_agents(self, world):
        return [agent for agent in world.agents if not agent.adversary]

    def adversaries(self, world):
        return [agent for agent in world.agents if agent.adversary]

    def is_collision(self, agent1, agent2):        # simple_tag.py:69-73 (strict <)
        d = agent1.state.p_pos - agent2.state.p_pos
        return torch.sqrt((d * d).sum(dim=1)) < (agent1.size + agent2.size)

    def benchmark_data(self, agent, world):        # simple_tag.py:57-66
        if agent.adversary:
            return sum(self.is_collision(a, agent).int() for a in self.good_agents(world))
        return torch.zeros(world.batch_size, dtype=torch.int32, device=world.device)

    def reward(self, agent, world):                # simple_tag.py:84-87
        return self.adversary_reward(agent, world) if agent.adversary else self.agent_reward(agent, world)

    def agent_reward(self, agent, world):          # simple_tag.py:89-113
        rew = torch.zeros(world.batch_size, dtype=torch.float32, device=world.device)
        if agent.collide:
            for a in self.adversaries(world):
                rew = rew - 10.0 * self.is_collision(a, agent).float()
        for p in range(world.dim_p):
            x = agent.state.p_pos[:, p].abs()
            far = torch.clamp(torch.exp(2 * x - 2), max=10.0)
            rew = rew - torch.where(x < 0.9, torch.zeros_like(x), torch.where(x < 1.0, (x - 0.9) * 10, far))
        return rew

    def adversary_reward(self, agent, world):      # simple_tag.py:115-129
        rew = torch.zeros(world.batch_size, dtype=torch.float32, device=world.device)
        if agent.collide:
            for ag in self.good_agents(world):
                for adv in self.adversaries(world):
                    rew = rew + 10.0 * self.is_collision(ag, adv).float()
        return rew

    def observation(self, agent, world):           # simple_tag.py:131-147
        entity_pos = [lm.state.p_pos - agent.state.p_pos for lm in world.landmarks if not lm.boundary]
        other_pos, other_vel = [], []
        for other in world.agents:
            if other is agent:
                continue
            other_pos.append(other.state.p_pos - agent.state.p_pos)
            if not other.adversary:
                other_vel.append(other.state.p_vel)
        return torch.cat([agent.state.p_vel, agent.state.p_pos] + entity_pos + other_pos + other_vel, dim=1)
