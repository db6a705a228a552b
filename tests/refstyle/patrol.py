"""A reference-STYLE scenario file (test fixture, written for this repo): a scripted agent, a pushable landmark, `done`.

patrol: two policy agents try to push a ball (a MOVABLE landmark, core.py:158-169 integrates it) onto a plate; a scripted
guard (`action_callback`, core.py:119-121) runs at the ball.  `done` when the ball rests on the plate; `benchmark_data`
returns a number.
"""
import numpy as np
from multiagent.core import World, Agent, Landmark, Action
from multiagent.scenario import BaseScenario


def chase_ball(agent, world):
    act = Action()
    d = world.landmarks[0].state.p_pos - agent.state.p_pos
    n = np.sqrt(np.sum(np.square(d)))
    act.u = (d / n if n > 1e-9 else np.zeros(world.dim_p)) * 1.5
    act.c = np.zeros(world.dim_c)
    return act


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.dim_c = 0
        world.agents = [Agent() for _ in range(3)]
        for i, agent in enumerate(world.agents):
            agent.name = "agent %d" % i
            agent.collide = True
            agent.silent = True
            agent.size = 0.08
            agent.max_speed = 1.2
        world.agents[2].name = "guard"
        world.agents[2].action_callback = chase_ball
        world.agents[2].size = 0.1
        ball, plate = Landmark(), Landmark()
        ball.name, ball.movable, ball.collide, ball.size, ball.initial_mass = "ball", True, True, 0.09, 0.6
        plate.name, plate.movable, plate.collide, plate.size = "plate", False, False, 0.3
        world.landmarks = [ball, plate]
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for agent in world.agents:
            agent.state.p_pos = np.random.uniform(-0.5, +0.5, world.dim_p)
            agent.state.p_vel = np.zeros(world.dim_p)
            agent.state.c = np.zeros(world.dim_c)
        for lm in world.landmarks:
            lm.state.p_pos = np.random.uniform(-0.5, +0.5, world.dim_p)
            lm.state.p_vel = np.zeros(world.dim_p)

    def gap(self, world):
        ball, plate = world.landmarks
        return np.sqrt(np.sum(np.square(ball.state.p_pos - plate.state.p_pos)))

    def reward(self, agent, world):
        ball = world.landmarks[0]
        near = np.sqrt(np.sum(np.square(agent.state.p_pos - ball.state.p_pos)))
        return -self.gap(world) - 0.1 * near

    def done(self, agent, world):
        return bool(self.gap(world) < world.landmarks[1].size)

    def benchmark_data(self, agent, world):
        return float(np.sqrt(np.sum(np.square(world.landmarks[0].state.p_vel))))

    def observation(self, agent, world):
        ball, plate = world.landmarks
        others = [o.state.p_pos - agent.state.p_pos for o in world.agents if o is not agent]
        return np.concatenate([agent.state.p_vel, agent.state.p_pos, ball.state.p_pos - agent.state.p_pos, ball.state.p_vel,
                               plate.state.p_pos - agent.state.p_pos] + others)
