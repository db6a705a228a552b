"""simple_speaker_listener: an immobile speaker sees the goal landmark's colour and talks; a silent
listener moves (reference: multiagent/scenarios/simple_speaker_listener.py).  Generic path."""
import torch

from .. import _abi
from ..core import World, Agent, Landmark, EntityChoice
from ..scenario import BaseScenario
from . import _util as U


class Scenario(BaseScenario):
    kind = _abi.MPE_SCN_SPEAKER_LISTENER   # fused kernel (wave-per-agent family)
    landmark_range = 1.0

    def make_world(self, batch_size=1, device=None):
        world = World(batch_size, device)          # simple_speaker_listener.py:6-32
        world.dim_c = 3
        num_landmarks = 3
        world.choice_pops = [num_landmarks]        # agents[0].goal_b = np.random.choice(world.landmarks), :41
        world.collaborative = True
        world.agents = [Agent() for _ in range(2)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
            agent.size = 0.075
        world.agents[0].movable = False   # speaker
        world.agents[1].silent = True     # listener
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.size = 0.04
        world.allocate()
        self._world = world
        self._apply(world)
        return world

    @property
    def goal_index(self):
        return self._world.choice_i32[0].long()

    def reset_world(self, world, mask=None, seeds=None):   # simple_speaker_listener.py:34-59
        idx = world.reset_uniform(self.landmark_range, mask, choices=[len(world.landmarks)], seeds=seeds)
        self.set_goal(world, World.merge_choice(self.goal_index, idx[:, 0], mask))

    def set_goal(self, world, index):
        world.choice_i32[0].copy_(torch.as_tensor(index, device=world.device).int())
        self._apply(world)

    def _apply(self, world):
        for agent in world.agents:
            agent.goal_a = None
            agent.goal_b = None
            if agent is not world.agents[1]:       # (the listener takes the goal's colour + 0.45 below, :52)
                agent.color = U.const(world, [0.25, 0.25, 0.25])
        world.landmarks[0].color = U.const(world, [0.65, 0.15, 0.15])
        world.landmarks[1].color = U.const(world, [0.15, 0.65, 0.15])
        world.landmarks[2].color = U.const(world, [0.15, 0.15, 0.65])
        world.agents[0].goal_a = world.agents[1]                                     # the listener ...
        world.agents[0].goal_b = EntityChoice(world, world.landmarks, world.choice_i32[0])   # ... should reach this landmark
        U.assign(world.agents[1], "color", world.agents[0].goal_b.color + 0.45)                  # :52 (rendering only)

    def benchmark_data(self, agent, world):
        # the reference's body (`self.reward(agent, reward)`, :61) raises NameError; this is what it means
        return self.reward(agent, world)

    def reward(self, agent, world):                # simple_speaker_listener.py:63-67
        a = world.agents[0]
        return -U.dist2(a.goal_a, a.goal_b)

    def observation(self, agent, world):           # simple_speaker_listener.py:69-92
        goal_color = U.zeros(world, world.dim_color)
        if agent.goal_b is not None:
            goal_color = agent.goal_b.color
        entity_pos = [entity.state.p_pos - agent.state.p_pos for entity in world.landmarks]
        comm = [other.state.c for other in world.agents if other is not agent and other.state.c is not None]
        if not agent.movable:      # speaker
            return torch.cat([goal_color], dim=1)
        if agent.silent:           # listener
            return torch.cat([agent.state.p_vel] + entity_pos + comm, dim=1)
