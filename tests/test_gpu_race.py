"""World.step forces come from the PRE-step positions of all entities (core.py:117-131).  In the wave-per-agent
kernel (csrc/mpe_split.hip) agent wave i stores its new state while sibling waves may still be loading the old
one; the stores therefore sit behind the workgroup barrier.  Two checks that do not depend on dispatch luck:
  * a test build in which one agent wave of every workgroup starts ~30 us late (libmpe_hip_stress.so) must give
    bit-identical results to the normal build -- and the same build with the round-1 ordering (stores in front
    of the barrier, libmpe_hip_stress_racy.so) must NOT: the negative control showing the probe sees the race;
  * split-vs-thread bit identity at 1 048 576 worlds over 100 steps.
The row-program kernel (csrc/mpe_rows.hip) shares its LDS between phases (the staged state, World.step's new state, the row tiles in
the same region, the next rollout step): the same two builds hold a different wave back at every phase boundary of a program step and
of a fused program rollout (examples/corral.py: colliding agents) -- identical results with the barriers, different ones without the
barrier between World.step and the new state taking the staged one's place.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _build

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def probe(lib, B=4096, steps=6):
    env = dict(os.environ)
    if lib:
        env["MPE_HIP_LIB"] = lib
    else:
        env.pop("MPE_HIP_LIB", None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_race_probe.py"), str(B), str(steps)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RACE_PROBE ")][-1]
    return json.loads(line[len("RACE_PROBE "):])


def test_delayed_agent_wave_changes_nothing_and_the_old_ordering_is_caught():
    for tag in _build.STRESS_VARIANTS:
        assert os.path.exists(_build.variant_lib(tag)), "build the stress variants: python -m multiagent_particle_envs_amd._build"
    normal = probe(None)
    delayed = probe(_build.variant_lib("stress"))
    assert delayed == normal, "a late agent wave changed the results: state stores are visible to sibling loads"
    racy = probe(_build.variant_lib("stress_racy"))
    for k in normal:   # every probed scenario has colliding agents: the old ordering must show
        assert racy[k] != normal[k], "negative control failed for %s: the probe cannot see the race" % k


def test_split_vs_thread_bit_identity_1M_worlds_100_steps():
    B = 1 << 20
    rs = np.random.RandomState(11)
    envs = {}
    for impl in ("split", "thread"):
        e = mpe.make_env("simple_spread", batch_size=B, seed=5)
        e.step_impl = impl
        envs[impl] = e
    A, E = 3, 6
    pos = rs.uniform(-1, 1, (B, E, 2)).astype(np.float32)
    pos[::2] *= 0.3
    vel = rs.uniform(-1, 1, (B, A, 2)).astype(np.float32)
    for e in envs.values():
        e.world.set_state(pos, vel)
    ids = torch.empty((16, A, B), dtype=torch.int32, device="cuda")
    for t in range(16):
        ids[t] = torch.randint(0, 5, (A, B), device="cuda", dtype=torch.int32)
    for e in envs.values():
        e.discrete_action_input = True
    for t in range(100):
        outs = {k: e.step(ids[t % 16]) for k, e in envs.items()}
        if t % 10 == 9 or t < 3:
            for a, b in zip(outs["split"][0] + outs["split"][1], outs["thread"][0] + outs["thread"][1]):
                assert torch.equal(a, b), "step %d" % t
            assert torch.equal(envs["split"].world.pos, envs["thread"].world.pos)
            assert torch.equal(envs["split"].world.vel, envs["thread"].world.vel)


def test_rollout_and_scenario_suites_pass_under_the_delayed_wave_build():
    """The fused rollouts double-buffer their LDS exchange blocks (and the moves the reward wave draws ahead) by step
    parity, with one barrier per step: exactly the kind of protocol a late wave would break.  Run the rollout and
    scenario and parity suites -- bit-identity of rollout vs stepwise for all nine scenarios and for k_duo_roll, k_duo vs
    k_wave, fused vs generic, goldens -- in a subprocess against libmpe_hip_stress.so (one wave of every workgroup of
    k_split / k_duo ~30 us late; in k_duo_roll a different wave lags at every step)."""
    env = dict(os.environ)
    env["MPE_HIP_LIB"] = _build.variant_lib("stress")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_rollout.py"),
                        os.path.join(ROOT, "tests", "test_f3_scenarios.py"), os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_server.py"),      # (the step server: the same loop, commanded)
                        "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


_NAV6 = '''
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.collaborative = True
        world.agents = [Agent() for _ in range(6)]
        for i, agent in enumerate(world.agents):
            agent.name, agent.collide, agent.silent, agent.size = "agent %d" % i, True, True, 0.15
        world.landmarks = [Landmark() for _ in range(6)]
        for i, lm in enumerate(world.landmarks):
            lm.name, lm.collide, lm.movable = "landmark %d" % i, False, False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for e in world.agents + world.landmarks:
            e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            e.state.p_vel = np.zeros(world.dim_p)
        for a in world.agents:
            a.state.c = np.zeros(world.dim_c)

    def reward(self, agent, world):
        rew = 0
        for lm in world.landmarks:
            rew -= min(np.linalg.norm(a.state.p_pos - lm.state.p_pos) for a in world.agents)
        for a in world.agents:
            if a is not agent and np.linalg.norm(a.state.p_pos - agent.state.p_pos) < a.size + agent.size:
                rew -= 1
        return rew

    def observation(self, agent, world):
        return np.concatenate([agent.state.p_vel, agent.state.p_pos] + [lm.state.p_pos - agent.state.p_pos for lm in world.landmarks])
'''


def test_shared_reward_values_wait_for_a_late_wave(tmp_path):
    """A traced program computes what its agents' rewards share once per world: the tasks are dealt to the workgroup's waves, the
    values cross through LDS, ONE barrier stands between the writers and the readers.  The program's image compiled with one wave
    of every workgroup held back ~30 us at the phase boundary in front of the shared tasks (MPE_STRESS_DELAY_WAVE) gives the same
    bits; the same build without that barrier does not."""
    path = tmp_path / "nav6.py"
    path.write_text(_NAV6)

    def run(flags):
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        env.pop("MPE_ROWS_IMAGE_FLAGS", None)
        if flags:
            env["MPE_ROWS_IMAGE_FLAGS"] = flags
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_race_probe_traced.py"), str(path), "4096", "4"],
                           capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("RACE_PROBE ")][-1][len("RACE_PROBE "):])
    normal = run(None)
    assert normal["n_shared"] == 6
    assert run("-DMPE_STRESS_DELAY_WAVE=1") == normal, "a late wave changed the results: shared values were read before they were written"
    assert run("-DMPE_STRESS_DELAY_WAVE=1 -DMPE_STRESS_NO_SHARED_BARRIER")["sha"] != normal["sha"], "negative control failed: the probe cannot see the race"
