"""simple_crypto: Alice (speaker) sends Bob a goal colour under a private key while Eve listens; nobody
moves (reference: multiagent/scenarios/simple_crypto.py).  Pure communication game; generic path (the
physics kernel has nothing to integrate: every agent is immovable)."""
import torch

from .. import _abi
from ..core import World, Agent, Landmark, EntityChoice
from ..scenario import BaseScenario
from . import _util as U


class CryptoAgent(Agent):
    def __init__(self):
        super(CryptoAgent, self).__init__()
        self.key = None


class Scenario(BaseScenario):
    kind = _abi.MPE_SCN_CRYPTO   # fused kernel (wave-per-agent family)
    num_adversaries = 1
    landmark_range = 1.0

    def make_world(self, batch_size=1, device=None):
        world = World(batch_size, device)          # simple_crypto.py:21-45
        num_agents, num_adversaries, num_landmarks = 3, 1, 2
        world.dim_c = 4
        world.choice_pops = [num_landmarks, num_landmarks]   # goal (:63), then the key (:65)
        world.agents = [CryptoAgent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
            agent.adversary = True if i < num_adversaries else False
            agent.speaker = True if i == 2 else False
            agent.movable = False
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        world.allocate()
        self._world = world
        self._apply(world)
        return world

    @property
    def choice_index(self):                        # [B, 2] long: goal landmark, key landmark
        return self._world.choice_i32.t().long()

    def reset_world(self, world, mask=None, seeds=None):   # simple_crypto.py:48-78: goal, then key, then positions
        n = len(world.landmarks)
        idx = world.reset_uniform(self.landmark_range, mask, choices=[n, n], seeds=seeds)
        m = None if mask is None else torch.as_tensor(mask, device=world.device).bool()[:, None]
        self.set_choices(world, World.merge_choice(self.choice_index, idx, m))

    def set_choices(self, world, index):
        """index [B, 2]: goal landmark, key landmark."""
        world.choice_i32.copy_(torch.as_tensor(index, device=world.device).reshape(world.batch_size, 2).t().int())
        self._apply(world)

    def _apply(self, world):
        for agent in world.agents:
            if agent is not world.agents[1]:       # (agents[1] takes the goal's colour below, :67)
                agent.color = U.const(world, [0.75, 0.25, 0.25] if agent.adversary else [0.25, 0.25, 0.25])
        for i, landmark in enumerate(world.landmarks):   # :56-61: landmark i's "colour" is the one-hot e_i of width dim_c
            c = [0.0] * world.dim_c
            c[i] += 1
            landmark.color = U.const(world, c)
        goal = EntityChoice(world, world.landmarks, world.choice_i32[0])
        U.assign(world.agents[1], "color", goal.color)
        U.assign(world.agents[2], "key", EntityChoice(world, world.landmarks, world.choice_i32[1]).color)
        for agent in world.agents:
            agent.goal_a = goal

    def benchmark_data(self, agent, world):        # simple_crypto.py:81-83
        return (agent.state.c, agent.goal_a.color)

    def good_listeners(self, world):
        return [agent for agent in world.agents if not agent.adversary and not agent.speaker]

    def good_agents(self, world):
        return [agent for agent in world.agents if not agent.adversary]

    def adversaries(self, world):
        return [agent for agent in world.agents if agent.adversary]

    def reward(self, agent, world):                # simple_crypto.py:97-98
        return self.adversary_reward(agent, world) if agent.adversary else self.agent_reward(agent, world)

    @staticmethod
    def _err(a, agent):
        """np.sum(np.square(a.state.c - goal.color)) unless a.state.c is all zeros (then 0: the `continue`)."""
        c = a.state.c
        e = ((c - agent.goal_a.color) ** 2).sum(dim=1)
        silent = (c == 0).all(dim=1)
        return torch.where(silent, torch.zeros_like(e), e)

    def agent_reward(self, agent, world):          # simple_crypto.py:100-117
        good_rew = U.zeros(world)
        adv_rew = U.zeros(world)
        for a in self.good_listeners(world):
            good_rew = good_rew - self._err(a, agent)
        for a in self.adversaries(world):
            adv_rew = adv_rew + self._err(a, agent)
        return adv_rew + good_rew

    def adversary_reward(self, agent, world):      # simple_crypto.py:119-124
        return U.zeros(world) - self._err(agent, agent)

    def observation(self, agent, world):           # simple_crypto.py:127-169
        goal_color = agent.goal_a.color if agent.goal_a is not None else U.zeros(world, world.dim_color)
        comm = [other.state.c for other in world.agents
                if other is not agent and other.state.c is not None and other.speaker]
        if world.agents[2].key is None:
            key = U.zeros(world, world.dim_c)
            goal_color = U.zeros(world, world.dim_c)
        else:
            key = world.agents[2].key
        if agent.speaker:
            return torch.cat([goal_color, key], dim=1)
        if not agent.speaker and not agent.adversary:
            return torch.cat([key] + comm, dim=1)
        return torch.cat(comm, dim=1)
