"""simple_push: keep-away -- one good agent is rewarded for reaching the goal landmark, one adversary for
pushing it away (reference: multiagent/scenarios/simple_push.py).  Generic path."""
import torch

from .. import _abi
from ..core import World, Agent, Landmark, EntityChoice
from ..scenario import BaseScenario
from . import _util as U


class Scenario(BaseScenario):
    kind = _abi.MPE_SCN_PUSH                       # fused kernel
    landmark_range = 1.0
    num_adversaries = 1

    def make_world(self, batch_size=1, device=None):
        world = World(batch_size, device)          # simple_push.py:6-32
        world.dim_c = 2
        num_agents, num_adversaries, num_landmarks = 2, 1, 2
        world.choice_pops = [num_landmarks]        # goal = np.random.choice(world.landmarks), :41
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.adversary = True if i < num_adversaries else False
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.index = i
        world.allocate()
        self._world = world
        self._apply(world)
        return world

    @property
    def goal_index(self):                          # [B] long; stored as world.choice_i32[0] (what the kernels read)
        return self._world.choice_i32[0].long()

    def reset_world(self, world, mask=None, seeds=None):   # simple_push.py:34-58
        idx = world.reset_uniform(self.landmark_range, mask, choices=[len(world.landmarks)], seeds=seeds)
        self.set_goal(world, World.merge_choice(self.goal_index, idx[:, 0], mask))

    def set_goal(self, world, index):
        world.choice_i32[0].copy_(torch.as_tensor(index, device=world.device).int())
        self._apply(world)

    def _apply(self, world):
        for i, landmark in enumerate(world.landmarks):   # :36-39  color = 0.1 grey, channel i+1 += 0.8
            c = [0.1, 0.1, 0.1]
            c[i + 1] += 0.8
            landmark.color = U.const(world, c)
        goal = EntityChoice(world, world.landmarks, world.choice_i32[0])
        for agent in world.agents:                        # :41-49
            agent.goal_a = goal
            if agent.adversary:
                agent.color = U.const(world, [0.75, 0.25, 0.25])
            else:                                          # channel goal.index + 1 += 0.5
                U.assign(agent, "color", U.const(world, [0.25, 0.25, 0.25]) + U.one_hot_rows(world, self.goal_index + 1, 3, 0.5))

    def reward(self, agent, world):                # simple_push.py:60-62
        return self.adversary_reward(agent, world) if agent.adversary else self.agent_reward(agent, world)

    def agent_reward(self, agent, world):          # :64-66
        return -U.dist(agent, agent.goal_a)

    def adversary_reward(self, agent, world):      # :68-76
        agent_dist = [U.dist(a, a.goal_a) for a in world.agents if not a.adversary]
        pos_rew = torch.stack(agent_dist).min(dim=0).values
        neg_rew = U.dist(agent.goal_a, agent)
        return pos_rew - neg_rew

    def observation(self, agent, world):           # simple_push.py:78-96
        entity_pos = [entity.state.p_pos - agent.state.p_pos for entity in world.landmarks]
        entity_color = [entity.color for entity in world.landmarks]
        other_pos = [other.state.p_pos - agent.state.p_pos for other in world.agents if other is not agent]
        if not agent.adversary:
            return torch.cat([agent.state.p_vel, agent.goal_a.state.p_pos - agent.state.p_pos, agent.color]
                             + entity_pos + entity_color + other_pos, dim=1)
        return torch.cat([agent.state.p_vel] + entity_pos + other_pos, dim=1)
