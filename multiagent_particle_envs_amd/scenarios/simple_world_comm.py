"""simple_world_comm: predator-prey with an obstacle, food, forests that hide whoever stands in them, and
a leader predator that can talk (reference: multiagent/scenarios/simple_world_comm.py).  Generic path."""
import torch

from .. import _abi
from ..core import World, Agent, Landmark
from ..scenario import BaseScenario
from . import _util as U


class Scenario(BaseScenario):
    kind = _abi.MPE_SCN_WORLD_COMM   # fused kernel (wave-per-agent family)
    num_adversaries = 4
    landmark_range = 0.9                           # simple_world_comm.py:104-113

    def make_world(self, batch_size=1, device=None, num_good_agents=2, num_adversaries=4):
        """The reference's world (2 prey, 4 predators) by default.  Its callbacks are written for any team sizes
        (good_agents / adversaries lists, simple_world_comm.py:126-289), so those are arguments here; the obstacle, the
        two food items and the two forests stay (the observation reads forests[0] and forests[1] by index, :241-248)."""
        world = World(batch_size, device)          # simple_world_comm.py:7-61
        world.dim_c = 4
        self.num_adversaries = num_adversaries
        num_agents = num_adversaries + num_good_agents
        num_landmarks, num_food, num_forests = 1, 2, 2
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.leader = True if i == 0 else False
            agent.silent = True if i > 0 else False
            agent.adversary = True if i < num_adversaries else False
            agent.size = 0.075 if agent.adversary else 0.045
            agent.accel = 3.0 if agent.adversary else 4.0
            agent.max_speed = 1.0 if agent.adversary else 1.3
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = True
            landmark.movable = False
            landmark.size = 0.2
            landmark.boundary = False
        world.food = [Landmark() for _ in range(num_food)]
        for i, landmark in enumerate(world.food):
            landmark.name = 'food %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.size = 0.03
            landmark.boundary = False
        world.forests = [Landmark() for _ in range(num_forests)]
        for i, landmark in enumerate(world.forests):
            landmark.name = 'forest %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.size = 0.3
            landmark.boundary = False
        world.landmarks += world.food
        world.landmarks += world.forests
        world.allocate()
        self._colors(world)
        return world

    def _colors(self, world):                       # simple_world_comm.py:91-101 (rendering only)
        for agent in world.agents:
            c = [0.45, 0.95, 0.45] if not agent.adversary else [0.95, 0.45, 0.45]
            if agent.leader:
                c = [x - 0.3 for x in c]
            agent.color = U.const(world, c)
        for landmark in world.landmarks:
            landmark.color = U.const(world, [0.25, 0.25, 0.25])
        for landmark in world.food:
            landmark.color = U.const(world, [0.15, 0.15, 0.65])
        for landmark in world.forests:
            landmark.color = U.const(world, [0.6, 0.9, 0.6])

    def reset_world(self, world, mask=None, seeds=None):   # simple_world_comm.py:89-113
        # the reference places every landmark, then food again, then the forests again (:104-113)
        ents = world.entities
        again = [ents.index(l) for l in world.food] + [ents.index(l) for l in world.forests]
        world.reset_uniform(self.landmark_range, mask, seeds=seeds, redraw=again)
        self._colors(world)

    def benchmark_data(self, agent, world):        # simple_world_comm.py:115-123
        if agent.adversary:
            return sum(U.is_collision(a, agent).int() for a in self.good_agents(world))
        return torch.zeros(world.batch_size, dtype=torch.int32, device=world.device)

    def is_collision(self, agent1, agent2):        # :126-130
        return U.is_collision(agent1, agent2)

    def good_agents(self, world):
        return [agent for agent in world.agents if not agent.adversary]

    def adversaries(self, world):
        return [agent for agent in world.agents if agent.adversary]

    def reward(self, agent, world):                # :143-147
        return self.adversary_reward(agent, world) if agent.adversary else self.agent_reward(agent, world)

    def agent_reward(self, agent, world):          # simple_world_comm.py:156-186
        rew = U.zeros(world)
        if agent.collide:
            for a in self.adversaries(world):
                rew = rew - 5.0 * U.is_collision(a, agent).float()
        for p in range(world.dim_p):
            rew = rew - 2 * U.bound(agent.state.p_pos[:, p].abs())
        for food in world.food:
            rew = rew + 2.0 * U.is_collision(agent, food).float()
        rew = rew + 0.05 * torch.stack([U.dist(food, agent) for food in world.food]).min(dim=0).values
        return rew

    def adversary_reward(self, agent, world):      # simple_world_comm.py:188-203
        rew = U.zeros(world)
        agents = self.good_agents(world)
        rew = rew - 0.1 * torch.stack([U.dist(a, agent) for a in agents]).min(dim=0).values
        if agent.collide:
            for ag in agents:
                for adv in self.adversaries(world):
                    rew = rew + 5.0 * U.is_collision(ag, adv).float()
        return rew

    def observation(self, agent, world):           # simple_world_comm.py:231-289
        entity_pos = [entity.state.p_pos - agent.state.p_pos for entity in world.landmarks if not entity.boundary]
        inf1 = U.is_collision(agent, world.forests[0])
        inf2 = U.is_collision(agent, world.forests[1])
        one = torch.ones((world.batch_size, 1), dtype=torch.float32, device=world.device)
        in_forest = [torch.where(inf1[:, None], one, -one), torch.where(inf2[:, None], one, -one)]
        other_pos, other_vel = [], []
        for other in world.agents:
            if other is agent:
                continue
            oth_f1 = U.is_collision(other, world.forests[0])
            oth_f2 = U.is_collision(other, world.forests[1])
            if agent.leader:
                visible = torch.ones_like(inf1)
            else:   # same forest, or both in the open (:253)
                visible = (inf1 & oth_f1) | (inf2 & oth_f2) | (~inf1 & ~oth_f1 & ~inf2 & ~oth_f2)
            v = visible[:, None].float()
            other_pos.append((other.state.p_pos - agent.state.p_pos) * v)
            if not other.adversary:
                other_vel.append(other.state.p_vel * v)
        comm = [world.agents[0].state.c]
        head = [agent.state.p_vel, agent.state.p_pos]
        if agent.adversary:   # leader and followers share the layout (:281-285)
            return torch.cat(head + entity_pos + other_pos + other_vel + in_forest + comm, dim=1)
        return torch.cat(head + entity_pos + other_pos + in_forest + other_vel, dim=1)
