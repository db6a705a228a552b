"""A USER scenario written against this package's protocol, twice over (the example of scenario.py / rowspec.py):

    obs_spec / reward_spec      the scenario DESCRIBES its rows and reward terms -> mpe_step_rows: one launch per step, interpreted,
    [/ done_spec]               or -- env.compile_program() -- compiled in; with `arena` set, agents that leave it are done and, under
                                auto_reset, their worlds restart inside the same launch (mpe_step_rows_episode)
    observation / reward        the same rows COMPUTED with torch ops on [B, .] views -> the generic path (~100 launches per step)
    [/ done]

`make_env("examples/corral.py", batch_size=B)` takes the specs (fused=False: the torch callbacks).  tests/test_rowspec.py holds the
two against each other; tools/rowspec_rate.py and bench.py (`extra.user_scenario`) time them.
"""
import torch

from multiagent_particle_envs_amd import rowspec
from multiagent_particle_envs_amd.core import World, Agent, Landmark
from multiagent_particle_envs_amd.scenario import BaseScenario


class Scenario(BaseScenario):
    """Two herders and a stray; three posts, one of them (picked per world) is the gate.  Rows and rewards as specs AND as
    torch callbacks (the generic path), so that the two can be held against each other."""

    landmark_range = 0.9       # reset_world places the posts in [-0.9, 0.9)^2 (read by the device-side resets too)
    device_reset = True        # reset_world IS world.reset_uniform(landmark_range, choices=choice_pops): finished worlds may be
                               # restarted by the device-side draw (in the step launch / a rollout) instead of a masked reset_world
    arena = None               # a bound (e.g. 0.95): an agent outside |x|, |y| <= arena is done (None: never, the reference's default)

    def make_world(self, batch_size=1, device=None):
        world = World(batch_size, device)
        world.dim_c = 0
        world.choice_pops = [3]
        world.agents = [Agent() for _ in range(3)]
        for i, a in enumerate(world.agents):
            a.name, a.silent, a.collide = "agent %d" % i, True, True
            a.size = 0.1 if i < 2 else 0.05
            a.accel = 3.0 if i < 2 else 4.5
            a.max_speed = 1.0 if i < 2 else 1.4
        world.landmarks = [Landmark() for _ in range(3)]
        for l in world.landmarks:
            l.collide, l.movable, l.size = False, False, 0.08
        world.allocate()
        return world

    def reset_world(self, world, mask=None, seeds=None):
        idx = world.reset_uniform(self.landmark_range, mask, choices=[3], seeds=seeds)
        if world.choice_i32 is not None:
            world.choice_i32[0].copy_(World.merge_choice(world.choice_i32[0].long(), idx[:, 0].to(world.device), mask).to(torch.int32))

    def obs_spec(self, agent, world):
        o = rowspec.ObsSpec(world, agent)
        o.vel().pos().rel_pick(0, world.landmarks).onehot(0, 3, 0.1, 0.9)
        for l in world.landmarks:
            o.rel(l)
        for a in world.agents:
            if a is not agent:
                o.rel(a).vel(a)
        return o.const(0.5)

    def reward_spec(self, agent, world):
        r = rowspec.RewardSpec(world, agent)
        stray = world.agents[2]
        r.dist2_pick(stray, 0, world.landmarks).sqrt().add(-1.0)             # the stray's distance to the gate
        r.min_dist2_from(agent, world.landmarks).add(-0.25)                  # squared distance to the nearest post
        for a in world.agents:
            if a is not agent:
                r.add_if_touching(a, agent, -3.0)
        r.bound(agent, 0).add(-1.0).bound(agent, 1).add(-1.0)
        return r

    def done_spec(self, agent, world):
        return rowspec.DoneSpec(world, agent).outside(agent, self.arena) if self.arena is not None else None

    # the same in torch (generic path)
    def done(self, agent, world):
        if self.arena is None:
            return torch.zeros(world.batch_size, dtype=torch.bool, device=world.device)
        return (agent.state.p_pos.abs() > self.arena).any(dim=1)

    def _gate(self, world):
        pos = torch.stack([l.state.p_pos for l in world.landmarks])           # [3, B, 2]
        g = world.choice_i32[0].long()
        return pos[g, torch.arange(world.batch_size, device=pos.device)]

    def observation(self, agent, world):
        from multiagent_particle_envs_amd.scenarios._util import one_hot_rows
        me = agent.state.p_pos
        cols = [agent.state.p_vel, me, self._gate(world) - me, one_hot_rows(world, world.choice_i32[0], 3, 0.8) + 0.1]
        cols += [l.state.p_pos - me for l in world.landmarks]
        for a in world.agents:
            if a is not agent:
                cols += [a.state.p_pos - me, a.state.p_vel]
        cols.append(torch.full((world.batch_size, 1), 0.5, device=world.device))
        return torch.cat(cols, dim=1)

    def reward(self, agent, world):
        from multiagent_particle_envs_amd.scenarios._util import bound, dist2, is_collision
        stray = world.agents[2]
        d = stray.state.p_pos - self._gate(world)
        rew = -torch.sqrt((d * d).sum(dim=1))
        rew = rew - 0.25 * torch.stack([dist2(agent, l) for l in world.landmarks]).min(dim=0).values
        for a in world.agents:
            if a is not agent:
                rew = rew - 3.0 * is_collision(a, agent).float()
        return rew - bound(agent.state.p_pos[:, 0].abs()) - bound(agent.state.p_pos[:, 1].abs())
