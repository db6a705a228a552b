"""simple_spread: N agents cover N landmarks, shared reward with a collision penalty
(reference: multiagent/scenarios/simple_spread.py; 3/3 hard-coded there, parameters here).
Fused kernel kind MPE_SCN_SPREAD (wave-per-agent + reward wave for N <= 6, wave-per-world above)."""
import torch

from .. import _abi
from ..core import World, Agent, Landmark
from ..scenario import BaseScenario


class Scenario(BaseScenario):
    kind = _abi.MPE_SCN_SPREAD
    landmark_range = 1.0

    def make_world(self, batch_size=1, device=None, num_agents=3, num_landmarks=None):
        world = World(batch_size, device)          # simple_spread.py:7-29
        world.dim_c = 2
        num_landmarks = num_agents if num_landmarks is None else num_landmarks
        world.collaborative = True
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.size = 0.15
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        world.allocate()
        return world

    def reset_world(self, world, mask=None, seeds=None):       # simple_spread.py:31-45
        world.reset_uniform(self.landmark_range, mask, seeds=seeds)

    @staticmethod
    def _dist(a, b):
        d = a.state.p_pos - b.state.p_pos
        return torch.sqrt((d * d).sum(dim=1))

    def is_collision(self, agent1, agent2):        # simple_spread.py:66-70 (strict <)
        return self._dist(agent1, agent2) < (agent1.size + agent2.size)

    def _landmark_mins(self, world):
        return [torch.stack([self._dist(a, l) for a in world.agents]).min(dim=0).values for l in world.landmarks]

    def reward(self, agent, world):                # simple_spread.py:72-82
        rew = torch.zeros(world.batch_size, dtype=torch.float32, device=world.device)
        for m in self._landmark_mins(world):
            rew = rew - m
        if agent.collide:
            for a in world.agents:
                rew = rew - self.is_collision(a, agent).float()
        return rew

    def benchmark_data(self, agent, world):        # simple_spread.py:47-63
        mins = self._landmark_mins(world)
        min_dists = sum(mins)
        occupied = sum((m < 0.1).int() for m in mins)
        collisions = sum(self.is_collision(a, agent).int() for a in world.agents) if agent.collide else 0
        rew = -min_dists - (collisions.float() if torch.is_tensor(collisions) else 0.0)
        return (rew, collisions, min_dists, occupied)

    def observation(self, agent, world):           # simple_spread.py:84-100
        entity_pos = [lm.state.p_pos - agent.state.p_pos for lm in world.landmarks]
        comm, other_pos = [], []
        for other in world.agents:
            if other is agent:
                continue
            comm.append(other.state.c if other.state.c is not None else
                        torch.zeros((world.batch_size, world.dim_c), dtype=torch.float32, device=world.device))
            other_pos.append(other.state.p_pos - agent.state.p_pos)
        return torch.cat([agent.state.p_vel, agent.state.p_pos] + entity_pos + other_pos + comm, dim=1)
