"""Reference-style Scenario files on the traced path (symtrace.py -> a compiled row program with code ops, csrc/mpe_rows.hip):
`make_env('<file>.py', batch_size=B)` steps in ONE launch, the file's NumPy callbacks compiled into the kernel.

  * the reference's own nine files: their COMMITTED traces (tests/golden/traced_*.json -- the GPU box has no reference tree;
    tests/test_symtrace.py holds them to the files and to the goldens on the CPU) run on the device against the goldens the
    reference's env recorded: seeded resets, per-world picks, every step's rows / rewards / state at 1e-5;
  * the fixture files of tests/refstyle/ traced here, against their reference-recorded goldens and against the HOST path
    (refstyle.py: the same file's callbacks per world) on the same worlds;
  * episode ends / rollouts of row programs carry over; a traced program without its image refuses to run.
"""
import json
import os

import numpy as np
import pytest
import torch

import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _abi, refstyle, symtrace
from multiagent_particle_envs_amd.rollout import RandomRollout

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = os.path.join(HERE, "refstyle")
GOLDEN = os.path.join(HERE, "golden")
NINE = ["simple", "simple_spread", "simple_tag", "simple_adversary", "simple_push", "simple_speaker_listener", "simple_reference",
        "simple_crypto", "simple_world_comm"]
TOL = 1e-5


def close(a, b, what, tol=TOL):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    assert np.all(err <= tol), "%s: max scaled err %.3e" % (what, float(err.max()))
    return float(err.max()) if err.size else 0.0


def actions_of(g, t, n, dev):
    if "act" in g:
        return [torch.as_tensor(g["act"][t][:, i], dtype=torch.float32).to(dev) for i in range(n)]
    return [torch.as_tensor(g["act%d" % i][t], dtype=torch.float32).to(dev) for i in range(n)]


def check_info(name, info, g, t, ok):
    """benchmark_data of the traced env (make_env(..., benchmark=True): one more launch per step) against what the reference's env
    recorded: integer outputs exactly (outside the decision-margin band), numbers at 1e-5."""
    vals = info["n"]
    if name == "simple_spread":
        for i, v in enumerate(vals):
            for k, key in enumerate(("info_rew", "info_collisions", "info_min_dists", "info_occupied")):
                if k in (1, 3):
                    assert v[k].dtype == torch.int32 and np.array_equal(v[k].cpu().numpy()[ok], g[key][t][:, i][ok].astype(np.int64)), (key, t, i)
                else:
                    close(v[k][torch.as_tensor(ok)], g[key][t][:, i][ok], "%s t=%d" % (key, t))
    elif name in ("simple_tag", "simple_world_comm"):
        for i, v in enumerate(vals):
            assert v.dtype == torch.int32 and np.array_equal(v.cpu().numpy()[ok], g["info_collisions"][t][:, i][ok].astype(np.int64)), (t, i)
    elif name == "simple_adversary":
        close(vals[0], g["info_adv"][t][:, 0], "info_adv t=%d" % t)
        for j in (1, 2):
            for k in range(3):
                close(vals[j][k], g["info_good"][t][:, j - 1, k], "info_good t=%d" % t)
    elif name == "herd":
        for i, v in enumerate(vals):
            close(v[0][torch.as_tensor(ok)], g["info0"][t][:, i][ok], "info rew")
            assert v[1].dtype == torch.int32 and np.array_equal(v[1].cpu().numpy()[ok], g["info1"][t][:, i][ok].astype(np.int64))
    else:
        return False
    return True


def replay(env, g, dev, name=None):
    """Seeded reset, then every recorded step teacher-forced from the reference's state."""
    n, W = env.n, g["rew"].shape[1]
    T = g["rew"].shape[0]
    none = torch.zeros(W, dtype=torch.bool, device=dev)
    obs = env.reset(seeds=[int(s) for s in g["seeds"]])
    pos, _ = env.world.get_state()
    same = np.all(np.abs(pos - g["pos0"]) <= 1e-6, axis=(1, 2))
    if "staged" in g:
        assert same[~g["staged"]].all() and (~g["staged"]).sum() >= W // 8, "the seeded reset does not reproduce the reference's"
        same = same & ~g["staged"]           # (a staged world may have kept its positions and been given velocities)
    else:
        assert same.sum() >= W // 2, "the seeded reset does not reproduce the reference's"
    if "choice" in g and g["choice"].shape[1]:
        assert np.array_equal(env.world.choice_i32.cpu().numpy().T, g["choice"])
    worst, masked = 0.0, 0.0
    for i in range(n):
        worst = max(worst, close(obs[i][torch.as_tensor(same)], g["obs_reset%d" % i][same], "obs_reset%d" % i))
    env.world.set_state(g["pos0"], g["vel0"])
    obs = env.reset(mask=none)                       # (nothing is reset: the rows of the staged state)
    for i in range(n):
        worst = max(worst, close(obs[i], g["obs_reset%d" % i], "rows of the staged reset state, agent %d" % i))
    for t in range(T):
        env.world.set_state(g["pos0"] if t == 0 else g["pos"][t - 1], g["vel0"] if t == 0 else g["vel"][t - 1])
        obs, rew, done, info = env.step(actions_of(g, t, n, dev))
        pos, vel = env.world.get_state()
        worst = max(worst, close(pos, g["pos"][t], "pos t=%d" % t), close(vel, g["vel"][t][:, :vel.shape[1]], "vel t=%d" % t))
        # fp32 kernel against the fp64 reference: worlds within 2e-6 of one of the file's own thresholds (contact tests ...) may
        # take the other branch; compared outside that band (found from the trace, on the reference's recorded state)
        tr = env.scenario.t
        V = np.zeros((W, tr.E, 2))
        V[:, :g["vel"][t].shape[1]] = g["vel"][t]
        Cw = np.zeros((W, tr.A, tr.dim_c))
        for i in range(tr.A):
            if tr.dim_c and "c%d" % i in g:
                Cw[:, i] = g["c%d" % i][t][:, :tr.dim_c]
        K = g["choice"].astype(np.int64) if "choice" in g and g["choice"].shape[1] else np.zeros((W, len(tr.pops)), np.int64)
        if getattr(tr, "params", ()):          # (random numbers the callbacks read: they sit in the env's pick slots since the seeded reset)
            K = env.world.choice_i32.cpu().numpy().T.astype(np.int64)
        roots = [x for row in tr.obs for x in row] + list(tr.rew)
        ok = symtrace.decision_margin(roots, W, P=g["pos"][t].astype(np.float64), V=V, Cw=Cw, K=K) > 2e-6
        assert ok.mean() >= 0.99, ok.mean()          # (the fused-kernel tests' bar: at most 1 % of the worlds on a knife edge)
        masked = max(masked, 1.0 - float(ok.mean()))
        okt = torch.as_tensor(ok)
        for i in range(n):
            worst = max(worst, close(obs[i][okt], g["obs%d" % i][t][ok], "obs%d t=%d" % (i, t)))
            worst = max(worst, close(rew[i][okt], g["rew"][t][:, i][ok], "rew%d t=%d" % (i, t)))
            assert not bool(done[i].any())
        if name is not None and tr.info is not None:
            ok_i = symtrace.decision_margin([x for row in tr.info for x in row], W, P=g["pos"][t].astype(np.float64), V=V, Cw=Cw, K=K) > 2e-6
            check_info(name, info, g, t, ok & ok_i)
            assert (ok & ok_i).mean() >= 0.99, (ok & ok_i).mean()
            masked = max(masked, 1.0 - float((ok & ok_i).mean()))
    replay.masked = masked
    return worst


@pytest.mark.parametrize("name", NINE)
def test_committed_traces_of_the_nine_reference_files_on_the_device(name, golden, record_parity):
    with open(os.path.join(GOLDEN, "traced_%s.json" % name)) as fh:
        data = json.load(fh)
    g = golden(name if name in ("simple", "simple_spread", "simple_tag") else "f3_" + name)
    W = g["rew"].shape[1]
    env = refstyle.make_traced_env(data, W, benchmark=True)
    assert env.traced and env.fused and env._prog.traced and env.program_compiled
    assert [s.shape[0] for s in env.observation_space] == [g["obs_reset%d" % i].shape[1] for i in range(env.n)]
    assert (env.info_callback is not None) == (name in ("simple_spread", "simple_tag", "simple_adversary", "simple_crypto", "simple_world_comm"))
    worst = replay(env, g, "cuda", name)
    record_parity("traced_" + name, {"what": "the reference's %s.py traced into the step kernel, against the goldens its own env recorded"
                                            % name, "worlds": W, "steps": int(g["rew"].shape[0]), "max_scaled_err": worst,
                                            "max_fraction_of_worlds_masked_knife_edge": replay.masked, "band": 2e-6})


@pytest.mark.parametrize("name", ["herd", "relay", "convoy", "survey", "mesh", "scatter"])
def test_fixture_files_traced_against_reference_goldens_and_the_host_path(name, golden, record_parity):
    path = os.path.join(FIXTURES, name + ".py")
    g = golden("refstyle_" + name)
    W = g["rew"].shape[1]
    env = mpe.make_env(path, batch_size=W, benchmark=True)
    assert env.traced and env.trace_fallback is None and env.program_compiled
    assert type(env.ref_scenario).__module__.startswith("mpe_user_scenario_")
    worst = replay(env, g, "cuda", name)
    record_parity("traced_fixture_" + name, {"what": "tests/refstyle/%s.py traced, against goldens recorded by the reference's env" % name,
                                             "worlds": W, "steps": int(g["rew"].shape[0]), "max_scaled_err": worst,
                                            "max_fraction_of_worlds_masked_knife_edge": replay.masked, "band": 2e-6})
    # ... and against the same file on the host path, free-running on more worlds (crowded at t = 3: contacts)
    B = 600
    a, b = mpe.make_env(path, batch_size=B, seed=5, benchmark=True), mpe.make_env(path, batch_size=B, seed=5, traced=False, benchmark=True)
    assert a.traced and not b.traced and not b.fused
    seeds = list(range(1000, 1000 + B))
    oa, ob = a.reset(seeds=seeds), b.reset(seeds=seeds)
    rs = np.random.RandomState(0)
    for t in range(8):
        if t == 3:
            p, v = a.world.get_state(all_entities=True)
            a.world.set_state(p * 0.25, v)
            b.world.set_state(p * 0.25, v)
        act = []
        for ag in a.agents:
            parts = ([np.eye(5, dtype=np.float32)[rs.randint(0, 5, B)]] if ag.movable else []) + \
                ([np.eye(a.world.dim_c, dtype=np.float32)[rs.randint(0, a.world.dim_c, B)]] if not ag.silent else [])
            act.append(torch.as_tensor(np.concatenate(parts, axis=1)).cuda())
        (oa, ra, _, ia), (ob, rb, _, ib) = a.step(act), b.step(act)
        assert np.array_equal(a.world.get_state()[0], b.world.get_state()[0])          # the same physics launch family: bit-identical
        # the host path evaluates the file in fp64, the kernel in fp32: a world within 2e-6 of one of the file's OWN thresholds
        # (a contact test, `gap < 0.25`) may take the other branch -- compared outside that band, as the contact counts of the
        # fused kernels are; the band is found from the trace itself
        tr = a.scenario.t
        P, V = a.world.get_state(all_entities=True)
        Cw = np.zeros((B, tr.A, tr.dim_c))
        if tr.dim_c:
            for i, ag in enumerate(a.world.agents):
                if not ag.silent:
                    Cw[:, i] = a._comm[i].cpu().numpy()
        K = a.world.choice_i32.cpu().numpy().T if tr.pops else np.zeros((B, 0), np.int64)
        roots = [n for row in tr.obs for n in row] + list(tr.rew)
        ok = symtrace.decision_margin(roots, B, P=P.astype(np.float64), V=V.astype(np.float64), Cw=Cw, K=K) > 2e-6
        assert ok.mean() > 0.9
        okt = torch.as_tensor(ok)
        for i in range(a.n):
            close(oa[i][okt], ob[i].cpu().numpy()[ok], "obs%d t=%d vs the host path" % (i, t))
            close(ra[i][okt], rb[i].cpu().numpy()[ok], "rew%d t=%d vs the host path" % (i, t))
        if tr.info is not None:      # benchmark_data in the file's own structure (herd: (reward, hits) per agent; hits as int32)
            ok_i = ok & (symtrace.decision_margin([x for row in tr.info for x in row], B, P=P.astype(np.float64), V=V.astype(np.float64), Cw=Cw, K=K) > 2e-6)
            for va, vb in zip(ia["n"], ib["n"]):
                assert isinstance(va, tuple) and len(va) == len(vb)
                for xa, xb in zip(va, vb):
                    assert xa.dtype == xb.dtype and xa.shape == xb.shape
                    close(xa[torch.as_tensor(ok_i)], xb.cpu().numpy()[ok_i], "benchmark_data t=%d vs the host path" % t)
        else:
            assert ia["n"] == ib["n"] == [{}] * a.n
    # ... and the check a user can run on live data: the file's own callbacks on the env's current state, a sample of worlds
    worst, checked = a.scenario.spot_check(a, oa, ra, worlds=120)
    assert worst <= TOL and checked >= 90, (worst, checked)


def test_a_traced_program_runs_compiled_in_only():
    env = mpe.make_env(os.path.join(FIXTURES, "relay.py"), batch_size=64)
    act = [torch.zeros((64, 4), device="cuda"), torch.zeros((64, 9), device="cuda")]
    env.reset()
    env.step(act)
    env._prog.unload()
    with pytest.raises(_abi.MpeError, match="compiled in only"):
        env.step(act)
    env.compile_program()
    env.step(act)
    # the file that cannot be traced still runs (host path), and says why
    env = mpe.make_env(os.path.join(FIXTURES, "patrol.py"), batch_size=4)
    assert not env.traced and "scripted agents" in env.trace_fallback
    with pytest.raises(symtrace.TraceUnsupported):
        mpe.make_env(os.path.join(FIXTURES, "patrol.py"), batch_size=4, traced=True)


def test_episode_ends_of_traced_envs():
    """relay's reset_world is World.reset_uniform's placement, herd places its agents in [-0.8, 0.8)^2 -- a box per entity: either
    way the episodes end and restart INSIDE the step launch, by the file's own placement."""
    W = 256
    rs = np.random.RandomState(1)
    for name, in_launch in (("relay", True), ("herd", True)):
        env = mpe.make_env(os.path.join(FIXTURES, name + ".py"), batch_size=W, max_episode_steps=3, auto_reset=True)
        assert env.traced and env._episode_in_launch == in_launch and env._device_restart_ok == in_launch
        env.reset()
        p0 = env.world.get_state()[0].copy()
        for t in range(1, 4):
            act = []
            for ag in env.agents:
                parts = ([np.eye(5, dtype=np.float32)[rs.randint(1, 5, W)]] if ag.movable else []) + \
                    ([np.eye(env.world.dim_c, dtype=np.float32)[rs.randint(0, env.world.dim_c, W)]] if not ag.silent else [])
                act.append(torch.as_tensor(np.concatenate(parts, axis=1)).cuda())
            obs, rew, done, _ = env.step(act)
            assert bool(done[0].all()) == (t == 3)
        pos, vel = env.world.get_state()
        assert np.all(vel == 0) and not np.allclose(pos, p0)              # every world restarted
        A = len(env.world.agents)
        assert np.abs(pos[:, :A]).max() < (0.8 if name == "herd" else 1.0)     # ... by ITS reset_world's placement
        if name == "herd":
            assert np.abs(pos[:, :A]).max() > 0.7 and np.abs(pos[:, A:]).max() > 0.9
        # the rows of the restarted worlds are the new episode's first: the agent's own position columns say so
        own = obs[1 if name == "herd" else 0]
        if name == "herd":
            assert np.allclose(own[:, 2:4].cpu().numpy(), pos[:, 1], atol=1e-6)


@pytest.mark.parametrize("name", ["relay", "herd"])
def test_fused_rollout_of_a_traced_env_equals_its_per_step_launches(name):
    """... whatever the placement of its resets: herd's per-entity boxes are drawn by mpe_reset_rows at the episode boundaries of
    the per-step form and inside the kernel in the fused form -- the same draws."""
    B, T = 2048, 12
    path = os.path.join(FIXTURES, name + ".py")
    a, b = mpe.make_env(path, batch_size=B, seed=9), mpe.make_env(path, batch_size=B, seed=9)
    ra, rb = RandomRollout(a, episode_len=5, pool=5, regenerate=True), RandomRollout(b, episode_len=5, pool=5, regenerate=True)
    ra.enqueue(T)
    rb.fused(T)
    torch.cuda.synchronize()
    assert torch.equal(a.world.pos, b.world.pos) and torch.equal(a.world.vel, b.world.vel)
    assert torch.equal(a.world.choice_i32, b.world.choice_i32)
    oa, ob = a._sets[(T - 1) & 1], b._sets[0]
    for x, y in zip(oa.obs_n, ob.obs_n):
        assert torch.equal(x, y)
    assert torch.equal(oa.rew, ob.rew)
    if name == "herd":
        A = len(a.world.agents)
        assert float(a.world.pos[:A].abs().max()) < 1.6 and a.scenario.reset_boxes(a.world) is not None


def test_a_reset_world_that_is_not_a_box_per_entity_is_evaluated_on_the_device_at_reset_time():
    """Positions that depend on a per-world pick (agents spawn on the side of the arena the pick names): no device-side draw fits;
    the traced reset program is drawn and evaluated with torch ops for all (masked) worlds at once, the episodes restart through
    the masked reset_callback, and a rollout that would reset on the device refuses with the reason."""
    from multiagent_particle_envs_amd import compat
    compat.install()
    from multiagent.core import World, Agent, Landmark
    from multiagent.scenario import BaseScenario

    class Sides(BaseScenario):
        def make_world(self):
            world = World()
            world.agents = [Agent() for _ in range(2)]
            for i, a in enumerate(world.agents):
                a.name, a.silent, a.size = "agent %d" % i, True, 0.05
            world.landmarks = [Landmark()]
            world.landmarks[0].name, world.landmarks[0].collide, world.landmarks[0].movable = "flag", False, False
            self.reset_world(world)
            return world

        def reset_world(self, world):
            world.side = np.random.choice([-1.0, 1.0])
            for a in world.agents:
                a.state.p_pos = np.random.uniform(-0.2, +0.2, world.dim_p) + np.array([0.6, 0.0]) * world.side
                a.state.p_vel = np.zeros(world.dim_p)
                a.state.c = np.zeros(world.dim_c)
            world.landmarks[0].state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            world.landmarks[0].state.p_vel = np.zeros(world.dim_p)

        def reward(self, agent, world):
            return -np.sum(np.square(agent.state.p_pos - world.landmarks[0].state.p_pos)) + 0.1 * world.side * agent.state.p_pos[0]

        def observation(self, agent, world):
            return np.concatenate([agent.state.p_vel, agent.state.p_pos, world.landmarks[0].state.p_pos - agent.state.p_pos, [world.side]])

    B = 1024
    env = refstyle.make_ref_env(Sides(), batch_size=B, seed=2, max_episode_steps=4, auto_reset=True)
    assert env.traced and not env.scenario.device_reset and not env._episode_in_launch
    obs = env.reset()
    side = obs[0][:, 6].cpu().numpy()
    x = env.world.get_state()[0][:, :2, 0]
    assert set(np.unique(side)) == {-1.0, 1.0} and np.all(np.abs(x - 0.6 * side[:, None]) < 0.2 + 1e-6)
    act = torch.zeros((2, B, 5), device="cuda")
    for t in range(1, 5):
        obs, rew, done, _ = env.step(act)
        assert bool(done[0].all()) == (t == 4)
    x = env.world.get_state()[0][:, :2, 0]
    side2 = obs[0][:, 6].cpu().numpy()
    assert np.all(np.abs(x - 0.6 * side2[:, None]) < 0.2 + 1e-6) and (side2 != side).mean() > 0.3      # restarted, by ITS placement
    with pytest.raises(_abi.MpeError, match="reset_uniform"):
        RandomRollout(refstyle.make_ref_env(Sides(), batch_size=64), episode_len=5)


def test_a_reset_world_that_cannot_be_traced_runs_on_the_host_at_reset_time_while_steps_stay_one_launch():
    """tests/refstyle/scatter.py: rejection sampling + normal draws in reset_world.  The env is traced (observation / reward in
    the step kernel), restarts go through the file's own reset_world per finished world; the spawn invariant of the file (nobody
    within 0.3 of anybody) holds after the first reset and after every automatic restart."""
    path = os.path.join(FIXTURES, "scatter.py")
    B = 512
    env = mpe.make_env(path, batch_size=B, seed=4, max_episode_steps=3, auto_reset=True)
    assert env.traced and env.scenario.t.host_reset and not env.scenario.device_reset and not env._episode_in_launch

    def clear(pos):
        gaps = np.linalg.norm(pos[:, :, None, :] - pos[:, None, :, :], axis=-1) + 10.0 * np.eye(pos.shape[1])
        return gaps.min(axis=(1, 2))
    env.reset()
    p0, v0 = env.world.get_state(all_entities=True)
    assert clear(p0).min() >= 0.3 - 1e-6 and 0.02 < np.abs(v0[:, :3]).mean() < 0.08 and np.all(v0[:, 3:] == 0)      # randn * 0.05
    act = [torch.as_tensor(np.eye(5, dtype=np.float32)[np.full(B, 1 + i)]).cuda() for i in range(env.n)]
    for t in range(1, 4):
        obs, rew, done, _ = env.step(act)
        assert bool(done[0].all()) == (t == 3)
    p1, v1 = env.world.get_state(all_entities=True)
    assert clear(p1).min() >= 0.3 - 1e-6 and (np.abs(p1 - p0).max(axis=(1, 2)) > 1e-3).mean() > 0.99          # restarted by ITS placement
    obs, rew, done, _ = env.step(act)          # (a step that ends no episode: rows and rewards are of the same state)
    worst, checked = env.scenario.spot_check(env, obs, rew, worlds=64)
    assert worst <= TOL and checked >= 40 and not bool(done[0].any())
    # the same seeds, the same worlds -- and another env seed, other worlds
    again = mpe.make_env(path, batch_size=B, seed=4)
    again.reset()
    assert np.array_equal(again.world.get_state(all_entities=True)[0], p0)
    other = mpe.make_env(path, batch_size=B, seed=5)
    other.reset()
    assert np.abs(other.world.get_state(all_entities=True)[0] - p0).max() > 0.1
    with pytest.raises(_abi.MpeError, match="reset_uniform"):
        RandomRollout(mpe.make_env(path, batch_size=64), episode_len=5)


def test_a_traced_done_callback_ends_episodes_inside_the_step_launch():
    """A reference-style Scenario with a `done(agent, world)` (the reference's done_callback, environment.py:132-135), asked for with
    done_callback=True: traced like the other callbacks, it becomes the program's done test -- with auto_reset the step, the test
    and the restart of the finished worlds are one launch -- and agrees with the host path's done rows."""
    from multiagent_particle_envs_amd import compat
    compat.install()
    from multiagent.core import World, Agent, Landmark
    from multiagent.scenario import BaseScenario

    class Fence(BaseScenario):
        def make_world(self):
            world = World()
            world.agents = [Agent() for _ in range(2)]
            for i, a in enumerate(world.agents):
                a.name, a.silent, a.size, a.accel = "agent %d" % i, True, 0.08, 4.0
            world.landmarks = [Landmark()]
            world.landmarks[0].name, world.landmarks[0].collide, world.landmarks[0].movable, world.landmarks[0].size = "post", False, False, 0.1
            self.reset_world(world)
            return world

        def reset_world(self, world):
            for a in world.agents:
                a.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
                a.state.p_vel = np.zeros(world.dim_p)
                a.state.c = np.zeros(world.dim_c)
            world.landmarks[0].state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            world.landmarks[0].state.p_vel = np.zeros(world.dim_p)

        def reward(self, agent, world):
            return -np.sqrt(np.sum(np.square(agent.state.p_pos - world.landmarks[0].state.p_pos)))

        def observation(self, agent, world):
            return np.concatenate([agent.state.p_vel, agent.state.p_pos, world.landmarks[0].state.p_pos - agent.state.p_pos])

        def done(self, agent, world):
            if abs(agent.state.p_pos[0]) > 1.05 or abs(agent.state.p_pos[1]) > 1.05:
                return True
            return bool(np.sqrt(np.sum(np.square(agent.state.p_pos - world.landmarks[0].state.p_pos))) < 0.12)

    B = 4096
    a = refstyle.make_ref_env(Fence(), batch_size=B, seed=1, done_callback=True)
    b = refstyle.make_ref_env(Fence(), batch_size=B, seed=1, done_callback=True, traced=False)
    assert a.traced and a._prog.has_done and not b.traced
    seeds = list(range(B))
    a.reset(seeds=seeds)
    b.reset(seeds=seeds)
    rs = np.random.RandomState(0)
    seen = 0
    for t in range(6):
        act = torch.as_tensor(np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(2, B))]).cuda()
        (_, _, da, _), (_, _, db, _) = a.step(act), b.step([act[0], act[1]])
        tr = a.scenario.t
        P, V = a.world.get_state(all_entities=True)
        ok = symtrace.decision_margin([d for d in tr.done], B, P=P.astype(np.float64), V=V.astype(np.float64), Cw=np.zeros((B, 2, 0)),
                                      K=np.zeros((B, 0), np.int64)) > 2e-6
        for i in range(2):
            assert np.array_equal(da[i].cpu().numpy()[ok], db[i].cpu().numpy()[ok]), (t, i)
            seen += int(da[i].sum())
    assert seen > 20                                                   # worlds did finish (walked out / reached the post)
    # with auto_reset the finished worlds restart inside the launch (the reset is World.reset_uniform's placement)
    c = refstyle.make_ref_env(Fence(), batch_size=B, seed=1, done_callback=True, max_episode_steps=50, auto_reset=True)
    assert c.traced and c._episode_in_launch
    c.reset(seeds=seeds)
    rs = np.random.RandomState(0)
    for t in range(6):
        act = torch.as_tensor(np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(2, B))]).cuda()
        _, _, dc, _ = c.step(act)
    assert int(c.episode_step.max()) == 6 and int((c.episode_step < 6).sum()) > 10      # some worlds restarted on the way


def test_every_node_kind_as_device_code_against_the_numpy_evaluation_of_the_trace():
    """A scenario that uses every kind of node the tracer knows (arithmetic, division, abs, min / max, sqrt, exp, log, tanh, sin, cos,
    atan2, comparisons and selects, picks, utterances, np.where / clip / maximum): the kernel's rows and rewards against the fp64
    NumPy evaluation of the same graphs on the kernel's own state."""
    from multiagent_particle_envs_amd import compat
    compat.install()
    from multiagent.core import World, Agent, Landmark
    from multiagent.scenario import BaseScenario

    class Zoo(BaseScenario):
        def make_world(self):
            world = World()
            world.dim_c = 3
            world.agents = [Agent() for _ in range(3)]
            for i, a in enumerate(world.agents):
                a.name, a.silent, a.size, a.max_speed = "agent %d" % i, i != 1, 0.06 + 0.02 * i, 1.2
            world.landmarks = [Landmark() for _ in range(2)]
            for l in world.landmarks:
                l.movable, l.collide, l.size = False, False, 0.1
            self.reset_world(world)
            return world

        def reset_world(self, world):
            world.mode = np.random.choice([0.5, 1.0, 2.0])
            world.target = np.random.choice(world.landmarks)
            for e in world.agents + world.landmarks:
                e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
                e.state.p_vel = np.zeros(world.dim_p)
            for a in world.agents:
                a.state.c = np.zeros(world.dim_c)

        def reward(self, agent, world):
            d = agent.state.p_pos - world.target.state.p_pos
            r = np.sqrt(np.sum(np.square(d)))
            speed = np.linalg.norm(agent.state.p_vel)
            ang = np.arctan2(d[1], d[0] + 1e-2)
            rew = -world.mode * r + 0.1 * np.cos(ang) - 0.05 * np.sin(2.0 * ang) + np.tanh(speed) / (1.0 + r)
            rew += 0.01 * np.log(1.0 + r * r) - np.exp(-3.0 * r) + max(abs(d[0]), abs(d[1])) * 0.2 - min(r, 0.7)
            rew -= np.sum(np.clip(np.abs(agent.state.p_pos) - 0.9, 0.0, 0.5))
            rew += 0.3 * np.sum(world.agents[1].state.c * np.array([1.0, -2.0, 0.5]))
            if r < agent.size + world.target.size:
                rew += 2.0
            elif r > 1.5:
                rew -= (r - 1.5) ** 2
            return rew

        def observation(self, agent, world):
            d = world.target.state.p_pos - agent.state.p_pos
            near = np.where(np.abs(d) < 0.5, d, np.sign(1.0) * 0.5 * np.ones(2))
            return np.concatenate([agent.state.p_vel / (1.0 + np.linalg.norm(agent.state.p_vel)), np.maximum(agent.state.p_pos, -0.5),
                                   d, near, [world.mode], world.agents[1].state.c, [np.exp(-np.dot(d, d))]])

    B = 8192
    env = refstyle.make_ref_env(Zoo(), batch_size=B, seed=4)
    assert env.traced, env.trace_fallback
    tr = env.scenario.t
    ops = set(n.op for n in symtrace.topo([x for row in tr.obs for x in row] + list(tr.rew)))
    assert {"add", "sub", "mul", "div", "abs", "min", "max", "sqrt", "exp", "log", "tanh", "sin", "cos", "atan2", "lt", "ite", "sel"} <= ops, ops
    env.reset()
    rs = np.random.RandomState(3)
    worst = 0.0
    for t in range(5):
        if t == 2:
            env.world.pos.mul_(0.3)
        moves = torch.as_tensor(np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(3, B))]).cuda()
        words = torch.as_tensor(rs.uniform(0, 1, size=(3, B, 3)).astype(np.float32)).cuda()
        obs, rew, _, _ = env.step((moves, words))
        P, V = env.world.get_state(all_entities=True)
        Cw = np.zeros((B, 3, 3))
        Cw[:, 1] = env._comm[1].cpu().numpy()
        K = env.world.choice_i32.cpu().numpy().T
        st = dict(P=P.astype(np.float64), V=V.astype(np.float64), Cw=Cw, K=K)
        roots = [x for row in tr.obs for x in row] + list(tr.rew)
        vals = symtrace.evaluate(roots, B, **st)
        ok = symtrace.decision_margin(roots, B, **st) > 2e-6
        assert ok.mean() > 0.97
        off = np.cumsum([0] + [len(r) for r in tr.obs])
        for i in range(3):
            want = np.stack(vals[off[i]:off[i + 1]], axis=1)
            worst = max(worst, close(obs[i][torch.as_tensor(ok)], want[ok], "obs%d t=%d" % (i, t)))
            worst = max(worst, close(rew[i][torch.as_tensor(ok)], vals[off[-1] + i][ok], "rew%d t=%d" % (i, t)))
    assert worst <= TOL


def test_traced_file_at_full_size_against_the_numpy_evaluation_and_shard_invariance():
    """tests/refstyle/convoy.py at BASELINE's 65 536 worlds: device resets + free-running steps, every row and reward against the
    fp64 NumPy evaluation of the trace on the kernel's own state (1e-5 outside the decision-margin band); and two shards of
    32 768 worlds numbered by their global world (`world_offset`) are, bit for bit, the one batch."""
    path = os.path.join(FIXTURES, "convoy.py")
    B = 65536
    env = mpe.make_env(path, batch_size=B, seed=6)
    halves = []
    for r in range(2):
        from multiagent_particle_envs_amd import scenarios
        ts = refstyle.trace_ref_scenario(scenarios.load(path).Scenario())
        w = ts.make_world(B // 2, None)
        w.seed, w.rng_mode, w.world_offset = 6, "device", r * (B // 2)
        ts.reset_world(w)
        h = mpe.MultiAgentEnv(w, ts.reset_world, None, None, fused=True, compile_program=True)
        h.scenario = ts
        halves.append(h)
    tr = env.scenario.t
    rs = np.random.RandomState(1)
    obs = env.reset()
    hobs = [h.reset() for h in halves]
    for i in range(env.n):
        assert torch.equal(obs[i], torch.cat([hobs[0][i], hobs[1][i]]))
    worst, masked = 0.0, 0.0
    for t in range(4):
        moves = torch.as_tensor(np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(env.n, B))]).cuda()
        obs, rew, _, _ = env.step(moves)
        houts = [h.step(moves[:, k * (B // 2):(k + 1) * (B // 2)].contiguous()) for k, h in enumerate(halves)]
        for i in range(env.n):
            assert torch.equal(obs[i], torch.cat([houts[0][0][i], houts[1][0][i]])) and torch.equal(rew[i], torch.cat([houts[0][1][i], houts[1][1][i]]))
        P, V = env.world.get_state(all_entities=True)
        st = dict(P=P.astype(np.float64), V=V.astype(np.float64), Cw=np.zeros((B, tr.A, tr.dim_c)), K=env.world.choice_i32.cpu().numpy().T)
        roots = [x for row in tr.obs for x in row] + list(tr.rew)
        vals = symtrace.evaluate(roots, B, **st)
        ok = symtrace.decision_margin(roots, B, **st) > 2e-6
        masked = max(masked, 1.0 - ok.mean())
        off = np.cumsum([0] + [len(r) for r in tr.obs])
        okt = torch.as_tensor(ok)
        for i in range(env.n):
            worst = max(worst, close(obs[i][okt], np.stack(vals[off[i]:off[i + 1]], axis=1)[ok], "obs%d t=%d" % (i, t)))
            worst = max(worst, close(rew[i][okt], vals[off[-1] + i][ok], "rew%d t=%d" % (i, t)))
    # (the first step after a reset masks ~2.5 % of the worlds: with the reference's constants -- contact force 100, dt 0.1, mass 1 --
    #  an entity that starts at rest overlapping an immovable one is pushed out by exactly its penetration, C dt^2 / m = 1, and lands
    #  ON the contact distance to the last bit: a genuine knife edge of the strict `<`; later steps mask ~3e-5)
    assert masked < 0.05 and worst <= TOL
