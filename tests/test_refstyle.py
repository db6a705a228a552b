"""Reference-STYLE Scenario files (the reference's own plug-in contract: `from multiagent.core import ...`, `make_world(self)`,
NumPy per-world callbacks -- multiagent/scenario.py:4-10, scenarios/__init__.py:5-7, make_env.py:36-43) loaded UNMODIFIED
through this package (compat/ import alias + refstyle.RefScenarioAdapter).

  -m gpu        the three fixture files of tests/refstyle/ (written for this repo against that contract) stepped by the HIP
                physics, against goldens the REFERENCE's own MultiAgentEnv recorded from the same files
                (tests/golden/gen_golden_refstyle.py): B worlds teacher-forced at 1e-5, and one world used exactly like the
                reference (NumPy in / out, np.random-seeded resets)
  not gpu       the host logic of the adapter with the physics step SUBSTITUTED BY THE ORACLE inside the test (the product
                has no CPU physics: `World.step` raises without a device) -- the same fixtures, and, in the build
                container, ALL NINE REFERENCE SCENARIO FILES loaded by path from /root/reference, unmodified, against the
                committed reference goldens (25 seeded, teacher-forced steps at 1e-5)
"""
import os
import sys

import numpy as np
import pytest
import torch

import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import compat, core, refstyle
from oracle import spec as ospec
from oracle.mpe_batched import BatchedOracle

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = os.path.join(HERE, "refstyle")
REF_SCENARIOS = "/root/reference/multiagent/scenarios"
TOL = 1e-5


def close(a, b, what="", tol=TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = np.maximum(1.0, np.abs(b))
    err = np.abs(a - b) / scale
    assert np.all(err <= tol), "%s: max scaled err %.3e at %s" % (what, float(err.max()), np.unravel_index(int(err.argmax()), err.shape))
    return float(err.max()) if err.size else 0.0


def np_(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


# ---- test-only physics: World.step of the device world answered by the fp64 oracle -------------------------------------------
def _oracle_world_step(self):
    """core.py:117-131 for the batched World with the oracle's arithmetic (TEST ONLY: lets the adapter's host logic run in a
    container without a GPU; the product's World.step is mpe_world_step or an error)."""
    for agent in self.scripted_agents:
        agent.action = agent.action_callback(agent, self)
    ents, A, B = self.entities, len(self.agents), self.batch_size
    sp = ospec.Spec(name="generic", n_agents=A, n_landmarks=len(self.landmarks), dim_c=int(self.dim_c),
                    size=[e.size for e in ents], movable=[bool(e.movable) for e in ents], collide=[bool(e.collide) for e in ents],
                    accel=[a.accel for a in self.agents], max_speed=[e.max_speed for e in ents], mass=[e.mass for e in ents],
                    dt=self.dt, damping=self.damping, contact_force=self.contact_force, contact_margin=self.contact_margin)
    orc = BatchedOracle(sp, B, np.float64)
    pos, vel = core.World.get_state(self, all_entities=True)
    orc.pos = pos.astype(np.float64)
    orc.vel = vel.astype(np.float64)[:, :orc.n_dyn]
    u = np.zeros((A, B, 2))
    for i, a in enumerate(self.agents):
        if a.movable and a.action.u is not None:
            u[i] = np_(self._as_batch(a.action.u, 2))
            assert not a.u_noise
    orc.integrate(orc.forces(u))
    v_all = np.zeros((B, len(ents), 2))
    v_all[:, :orc.n_dyn] = orc.vel
    core.World.set_state(self, orc.pos, v_all)
    for a in self.agents:
        a.state.c = torch.zeros((B, self.dim_c)) if a.silent else a.action.c


@pytest.fixture
def oracle_physics(monkeypatch):
    monkeypatch.setattr(core.World, "step", _oracle_world_step)
    monkeypatch.setattr(core.World, "_require_device", lambda self: None)


# ---- shared driver -------------------------------------------------------------------------------------------------------------
def actions_of(g, t, n, dev):
    if "act" in g:        # simple / simple_spread / simple_tag goldens: [T, W, A, 5]
        return [torch.as_tensor(g["act"][t][:, i], dtype=torch.float32).to(dev) for i in range(n)]
    return [torch.as_tensor(g["act%d" % i][t], dtype=torch.float32).to(dev) for i in range(n)]


def replay(env, g, dev, check_info=None, steps=None):
    """Seeded reset, then every recorded step teacher-forced from the reference's state: observations, rewards, dones, state."""
    n = env.n
    T, W = g["rew"].shape[:2]
    T = T if steps is None else min(T, steps)
    worst = 0.0
    obs = env.reset(seeds=[int(s) for s in g["seeds"]])
    pos, _ = env.world.get_state()
    same = np.all(np.abs(pos - g["pos0"]) <= 1e-6, axis=(1, 2))          # worlds the recorder did not squeeze / stage after the reset
    if "staged" in g:
        assert same[~g["staged"]].all() and (~g["staged"]).sum() >= W // 8, "the seeded reset does not reproduce the reference's"
        same = same & ~g["staged"]           # (a staged world may have kept its positions and been given velocities)
    else:
        assert same.sum() >= W // 2, "the seeded reset does not reproduce the reference's"
    for i in range(n):
        worst = max(worst, close(np_(obs[i])[same], g["obs_reset%d" % i][same], "obs_reset%d" % i))
    env.world.set_state(g["pos0"], g["vel0"])
    for i in range(n):
        worst = max(worst, close(np_(env._get_obs(env.agents[i])), g["obs_reset%d" % i], "obs of the staged reset state, agent %d" % i))
    for t in range(T):
        env.world.set_state(g["pos0"] if t == 0 else g["pos"][t - 1], g["vel0"] if t == 0 else g["vel"][t - 1])
        obs, rew, done, info = env.step(actions_of(g, t, n, dev))
        pos, vel = env.world.get_state()
        worst = max(worst, close(pos, g["pos"][t], "pos t=%d" % t), close(vel, g["vel"][t][:, :vel.shape[1]], "vel t=%d" % t))
        for i in range(n):
            worst = max(worst, close(np_(obs[i]), g["obs%d" % i][t], "obs%d t=%d" % (i, t)))
            worst = max(worst, close(np_(rew[i]) * np.ones(W), g["rew"][t][:, i], "rew%d t=%d" % (i, t)))
            if "done" in g:
                assert np.array_equal(np_(done[i]).astype(bool) | np.zeros(W, bool), g["done"][t][:, i]), ("done", t, i)
        if check_info:
            check_info(info, t)
    return worst


# ---- the import alias ----------------------------------------------------------------------------------------------------------
def test_alias_resolves_the_two_imports_a_scenario_file_makes():
    compat.install()
    from multiagent.core import World, Agent, Landmark
    from multiagent.scenario import BaseScenario
    w = World()
    assert (w.dim_c, w.dim_p, w.dt, w.damping, w.contact_force, w.contact_margin) == (0, 2, 0.1, 0.25, 100.0, 1e-3)   # core.py:89-99
    a, l = Agent(), Landmark()
    assert (a.movable, a.silent, a.collide, a.size, a.mass, a.u_range, a.max_speed, a.accel, a.action_callback) == \
        (True, False, True, 0.05, 1.0, 1.0, None, None, None)                                                        # core.py:27-79
    assert (l.movable, l.collide, l.size) == (False, True, 0.05) and a.state.c is None and a.action.u is None
    w.agents, w.landmarks = [a, Agent()], [l]
    w.agents[1].action_callback = lambda ag, wo: None
    assert w.entities == [a, w.agents[1], l] and w.policy_agents == [a] and w.scripted_agents == [w.agents[1]]    # core.py:101-114
    a2 = Agent()
    assert a2.state is not a.state and a2.action is not a.action          # no shared mutable defaults
    with pytest.raises(NotImplementedError):
        BaseScenario().make_world()
    with pytest.raises(RuntimeError, match="no CPU physics"):
        w.step()


def test_loader_tells_the_two_contracts_apart():
    sc = mpe.scenarios.load(os.path.join(FIXTURES, "herd.py")).Scenario()
    assert refstyle.is_reference_style(sc)
    assert not refstyle.is_reference_style(mpe.scenarios.load("simple_spread.py").Scenario())
    with pytest.raises(TypeError):
        mpe.make_env(os.path.join(FIXTURES, "herd.py"), batch_size=2, device="cpu", num_agents=5)


FIX = ["herd", "relay", "patrol", "convoy", "survey", "mesh", "scatter"]


def _info_checker(name, g, W):
    def chk(info, t):
        vals = info["n"]
        if name == "herd":           # benchmark_data -> (reward, hits): a tuple of two [B] tensors per agent
            for i, v in enumerate(vals):
                close(np_(v[0]), g["info0"][t][:, i], "info rew")
                assert np.array_equal(np_(v[1]).astype(np.int64), g["info1"][t][:, i].astype(np.int64))
        elif name == "patrol":       # a number per agent -> a [B] tensor
            for i, v in enumerate(vals):
                close(np_(v), g["info0"][t][:, i], "info ball speed")
        else:
            assert vals == [{} for _ in vals]
    return chk


def _make_fixture_env(name, W, dev):
    env = mpe.make_env(os.path.join(FIXTURES, name + ".py"), benchmark=True, batch_size=W, device=dev, traced=False)
    if name == "patrol":            # the recorder passed Scenario.done as done_callback (the reference's make_env passes none)
        env.done_callback = env.scenario.done
    return env


@pytest.mark.parametrize("name", FIX)
def test_fixture_files_host_logic_against_reference_goldens(name, golden, oracle_physics):
    g = golden("refstyle_" + name)
    W = g["rew"].shape[1]
    env = _make_fixture_env(name, W, "cpu")
    assert not env.fused and len(env.ref_worlds) == W and env.n == g["rew"].shape[2]
    replay(env, g, "cpu", _info_checker(name, g, W))


NINE = ["simple", "simple_spread", "simple_tag", "simple_adversary", "simple_push", "simple_speaker_listener", "simple_reference",
        "simple_crypto", "simple_world_comm"]


@pytest.mark.skipif(not os.path.isdir(REF_SCENARIOS), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("name", NINE)
def test_the_nine_reference_scenario_files_load_unmodified(name, golden, oracle_physics):
    """make_env('<reference>/multiagent/scenarios/<name>.py') -- the file as shipped, loaded by path -- against the goldens the
    reference itself recorded (tests/golden/gen_golden*.py): seeded resets, per-world picks (goal landmarks, keys, colours),
    ragged observations, shared rewards, communication, all through the file's own NumPy callbacks."""
    g = golden(name if name in ("simple", "simple_spread", "simple_tag") else "f3_" + name)
    W = g["rew"].shape[1]
    path = os.path.join(REF_SCENARIOS, name + ".py")
    before = open(path, "rb").read()
    env = mpe.make_env(path, batch_size=W, device="cpu")
    assert compat.installed() or "multiagent.core" in sys.modules
    assert type(env.ref_scenario).__module__.startswith("mpe_user_scenario_") and not env.fused
    ref_dims = [g["obs%d" % i].shape[-1] for i in range(env.n)]
    assert [s.shape[0] for s in env.observation_space] == ref_dims
    worst = replay(env, g, "cpu", steps=25)
    assert open(path, "rb").read() == before and worst <= TOL


@pytest.mark.skipif(not os.path.isdir(REF_SCENARIOS), reason="the reference tree exists in the build container only")
def test_one_world_numpy_io_is_the_reference_usage(golden, oracle_physics):
    """batch_size=None: ONE world, NumPy in / NumPy out, the process-global np.random stream consumed as the reference does
    (make_world's reset_world at construction, then reset()): world w of the golden, free-running from its seed."""
    g = golden("f3_simple_adversary")
    env = mpe.make_env(os.path.join(REF_SCENARIOS, "simple_adversary.py"))
    assert env.numpy_io and env.batch_size == 1
    w = int(np.flatnonzero(~g["staged"])[0])
    np.random.seed(int(g["seeds"][w]))
    obs = env.reset()
    assert isinstance(obs[0], np.ndarray) and obs[0].dtype == np.float64 and obs[0].ndim == 1
    for i in range(env.n):
        close(obs[i], g["obs_reset%d" % i][w], "reset obs", 1e-6)
    goal = env.ref_worlds[0].agents[0].goal_a                      # the file's own per-world Python state is there to look at
    assert env.ref_worlds[0].landmarks.index(goal) == int(g["choice"][w][0])
    for t in range(6):
        obs, rew, done, info = env.step([g["act%d" % i][t][w] for i in range(env.n)])
        assert isinstance(rew[0], float) and done == [False] * env.n and info == {"n": [{}] * env.n}
        for i in range(env.n):
            close(obs[i], g["obs%d" % i][t][w], "obs%d t=%d" % (i, t), 1e-4 if t else TOL)        # free-running: fp32 state drift
            close(rew[i], g["rew"][t][w, i], "rew t=%d" % t, 1e-4 if t else TOL)


def test_per_world_constants_are_refused(oracle_physics):
    """B worlds step in one launch and share their physics constants: a make_world that randomises one is an error, not a
    silently wrong batch."""
    compat.install()
    from multiagent.core import World, Agent, Landmark
    from multiagent.scenario import BaseScenario
    counter = [0]

    class Sc(BaseScenario):
        def make_world(self):
            w = World()
            w.agents = [Agent()]
            w.agents[0].silent = True
            w.agents[0].size = 0.05 + 0.01 * counter[0]
            counter[0] += 1
            w.landmarks = [Landmark()]
            self.reset_world(w)
            return w

        def reset_world(self, w):
            for e in w.entities:
                e.state.p_pos, e.state.p_vel = np.zeros(2), np.zeros(2)
            w.agents[0].state.c = np.zeros(w.dim_c)

        def reward(self, a, w):
            return 0.0

        def observation(self, a, w):
            return a.state.p_pos
    with pytest.raises(mpe._abi.MpeError, match="physics"):
        refstyle.make_ref_env(Sc(), batch_size=3, device="cpu")
    counter[0] = 0
    env = refstyle.make_ref_env(Sc(), batch_size=1, device="cpu")
    assert env.n == 1


# ---- on the GPU: the HIP physics under the same files -----------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", FIX)
def test_fixture_files_on_the_device_against_reference_goldens(name, golden, record_parity):
    g = golden("refstyle_" + name)
    W = g["rew"].shape[1]
    env = _make_fixture_env(name, W, "cuda")
    assert env.world.pos.is_cuda and not env.fused
    worst = replay(env, g, "cuda", _info_checker(name, g, W))
    record_parity("refstyle_%s" % name, {"worlds": W, "steps": int(g["rew"].shape[0]), "max_scaled_err": worst,
                                         "against": "tests/golden/refstyle_%s.npz: the reference's MultiAgentEnv stepping tests/refstyle/%s.py" % (name, name)})


@pytest.mark.gpu
def test_fixture_file_one_world_numpy_io_on_the_device(golden):
    g = golden("refstyle_herd")
    env = mpe.make_env(os.path.join(FIXTURES, "herd.py"), benchmark=True)
    assert env.numpy_io and env.world.pos.is_cuda
    for w in (0, 1, 3):                                 # worlds the recorder did not squeeze
        np.random.seed(int(g["seeds"][w]))
        obs = env.reset()
        for i in range(env.n):
            close(obs[i], g["obs_reset%d" % i][w], "reset obs", 1e-6)
        for t in range(5):
            obs, rew, done, info = env.step([g["act%d" % i][t][w] for i in range(env.n)])
            for i in range(env.n):
                close(obs[i], g["obs%d" % i][t][w], "obs%d t=%d" % (i, t), 2e-4 if t else TOL)
                close(rew[i], g["rew"][t][w, i], "rew", 2e-4 if t else TOL)
                assert isinstance(info["n"][i], tuple) and info["n"][i][1] == int(g["info1"][t][w, i])


@pytest.mark.gpu
def test_auto_reset_runs_the_files_reset_world_for_finished_worlds():
    """max_episode_steps + auto_reset on a reference-style env: the worlds that reach the horizon are reset by the FILE's
    reset_world (masked), the others keep stepping."""
    W = 8
    env = mpe.make_env(os.path.join(FIXTURES, "relay.py"), batch_size=W, max_episode_steps=3, auto_reset=True, traced=False)
    env.reset(seeds=list(range(W)))
    targets = [w.target for w in env.ref_worlds]
    rs = np.random.RandomState(0)
    for t in range(1, 4):
        act = [torch.as_tensor(np.eye(4, dtype=np.float32)[rs.randint(0, 4, W)]).cuda(),
               torch.as_tensor(np.concatenate([np.eye(5, dtype=np.float32)[rs.randint(0, 5, W)],
                                               np.eye(4, dtype=np.float32)[rs.randint(0, 4, W)]], axis=1)).cuda()]
        np.random.seed(1234)
        obs, rew, done, _ = env.step(act)
        assert bool(done[0].all()) == (t == 3)
    vel = env.world.get_state()[1]
    assert np.all(vel == 0)                                        # every world restarted: reset_world zeroes the velocities
    assert any(w.target != t0 for w, t0 in zip(env.ref_worlds, targets)) or W < 4
