"""simple_adversary: physical deception -- good agents cover the landmarks, one adversary has to find the
goal landmark by watching them (reference: multiagent/scenarios/simple_adversary.py).
Generic path: these callbacks are torch ops on [B, .] views; the physics is `mpe_world_step`."""
import torch

from .. import _abi
from ..core import World, Agent, Landmark, EntityChoice
from ..scenario import BaseScenario
from . import _util as U


class Scenario(BaseScenario):
    kind = _abi.MPE_SCN_ADVERSARY                  # fused kernel for the reference shape (3 agents, 1 adversary)
    landmark_range = 1.0
    num_adversaries = 1

    def make_world(self, batch_size=1, device=None, num_agents=3, num_adversaries=1):
        world = World(batch_size, device)          # simple_adversary.py:8-33
        world.dim_c = 2
        world.num_agents = num_agents
        self.num_adversaries = num_adversaries
        num_landmarks = num_agents - 1
        world.choice_pops = [num_landmarks]        # goal = np.random.choice(world.landmarks), :44
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
            agent.silent = True
            agent.adversary = True if i < num_adversaries else False
            agent.size = 0.15
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.size = 0.08
        world.allocate()
        self._world = world
        self._apply(world)
        return world

    @property
    def goal_index(self):                          # [B] long; stored as world.choice_i32[0] (what the kernels read)
        return self._world.choice_i32[0].long()

    def reset_world(self, world, mask=None, seeds=None):   # simple_adversary.py:35-55
        idx = world.reset_uniform(self.landmark_range, mask, choices=[len(world.landmarks)], seeds=seeds)
        self.set_goal(world, World.merge_choice(self.goal_index, idx[:, 0], mask))

    def set_goal(self, world, index):
        """goal = np.random.choice(world.landmarks) (:44) for every world: index [B]."""
        world.choice_i32[0].copy_(torch.as_tensor(index, device=world.device).int())
        self._apply(world)

    def _apply(self, world):
        world.agents[0].color = U.const(world, [0.85, 0.35, 0.35])
        for i in range(1, world.num_agents):
            world.agents[i].color = U.const(world, [0.35, 0.35, 0.85])
        for i, landmark in enumerate(world.landmarks):   # goal landmark green, the others grey (:41-46)
            is_goal = (self.goal_index == i).unsqueeze(1)
            U.assign(landmark, "color", torch.where(is_goal, U.const(world, [0.15, 0.65, 0.15]), U.const(world, [0.15, 0.15, 0.15])))
        goal = EntityChoice(world, world.landmarks, world.choice_i32[0])
        for agent in world.agents:
            agent.goal_a = goal

    def benchmark_data(self, agent, world):        # simple_adversary.py:57-66
        if agent.adversary:
            return U.dist2(agent, agent.goal_a)
        dists = [U.dist2(agent, l) for l in world.landmarks]
        dists.append(U.dist2(agent, agent.goal_a))
        return tuple(dists)

    def good_agents(self, world):
        return [agent for agent in world.agents if not agent.adversary]

    def adversaries(self, world):
        return [agent for agent in world.agents if agent.adversary]

    def reward(self, agent, world):                # simple_adversary.py:76-78
        return self.adversary_reward(agent, world) if agent.adversary else self.agent_reward(agent, world)

    def agent_reward(self, agent, world):          # :80-107 (shaped_reward = shaped_adv_reward = True)
        adv_rew = sum(U.dist(a, a.goal_a) for a in self.adversaries(world))
        pos_rew = -torch.stack([U.dist(a, a.goal_a) for a in self.good_agents(world)]).min(dim=0).values
        return pos_rew + adv_rew

    def adversary_reward(self, agent, world):      # :109-118 (shaped)
        return -U.dist2(agent, agent.goal_a)

    def observation(self, agent, world):           # simple_adversary.py:121-139
        entity_pos = [entity.state.p_pos - agent.state.p_pos for entity in world.landmarks]
        other_pos = [other.state.p_pos - agent.state.p_pos for other in world.agents if other is not agent]
        if not agent.adversary:
            return torch.cat([agent.goal_a.state.p_pos - agent.state.p_pos] + entity_pos + other_pos, dim=1)
        return torch.cat(entity_pos + other_pos, dim=1)
