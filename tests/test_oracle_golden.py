"""Pin the oracle to the reference: oracle/ vs the golden vectors recorded from /root/reference.

The reference ships no tests or fixtures; tests/golden/*.npz were produced
by tests/golden/gen_golden.py from the unmodified reference in the build container.
fp64 oracle paths must reproduce them to <=1e-12 (they are the same arithmetic in the same
order); integer outputs and done flags must be identical.
"""
import numpy as np
import pytest

from oracle import spec as ospec
from oracle.mpe_loop import LoopEnv
from oracle.mpe_batched import BatchedOracle, seeded_initial_state

CASES = [
    ("simple", ospec.simple(), False),
    ("simple_spread", ospec.simple_spread(3), True),
    ("simple_tag", ospec.simple_tag(), True),
    ("simple_spread_n5", ospec.simple_spread(5), True),
    ("simple_spread_n64", ospec.simple_spread(64), True),
]
TOL = 1e-12


def _close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    scale = np.maximum(1.0, np.abs(b))
    assert np.all(np.abs(a - b) <= tol * scale), float(np.max(np.abs(a - b) / scale))


@pytest.mark.parametrize("name,spec,bench", CASES, ids=[c[0] for c in CASES])
def test_batched_fp64_matches_reference(name, spec, bench, golden):
    g = golden(name)
    T, W, A = g["rew"].shape
    assert spec.obs_dims() == [g["obs%d" % i].shape[-1] for i in range(A)]
    orc = BatchedOracle(spec, W, np.float64, benchmark=bench)
    orc.set_state(g["pos0"], g["vel0"])
    for i, o in enumerate(orc.observe()):
        _close(o, g["obs_reset%d" % i])
    for t in range(T):
        obs, rew, done, info = orc.step(np.transpose(g["act"][t], (1, 0, 2)))
        _close(orc.pos, g["pos"][t])
        _close(orc.vel, g["vel"][t])
        for i in range(A):
            _close(obs[i], g["obs%d" % i][t])
        _close(rew.T, g["rew"][t])
        assert not done.any() and not g["done"][t].any()
        if "info_collisions" in g:
            assert np.array_equal(info["collisions"].T, g["info_collisions"][t])
        if "info_occupied" in g:
            assert np.array_equal(info["occupied_landmarks"].T, g["info_occupied"][t])
            _close(info["min_dists"].T, g["info_min_dists"][t])
            _close(info["rew"].T, g["info_rew"][t])


def test_batched_integer_action_ids(golden):
    g = golden("simple_spread_ids")
    T, W, A = g["rew"].shape
    orc = BatchedOracle(ospec.simple_spread(3), W, np.float64, benchmark=True)
    orc.set_state(g["pos0"], g["vel0"])
    for t in range(T):
        obs, rew, done, info = orc.step(ids=g["ids"][t].T)
        _close(orc.pos, g["pos"][t])
        for i in range(A):
            _close(obs[i], g["obs%d" % i][t])
        _close(rew.T, g["rew"][t])
        assert np.array_equal(info["collisions"].T, g["info_collisions"][t])


def test_seeded_reset_matches_reference(golden):
    for name, spec in [("simple", ospec.simple()), ("simple_tag", ospec.simple_tag())]:
        g = golden(name)
        W = len(g["seeds"])
        plain = [w for w in range(W) if name == "simple" or w % 3 != 2]  # squeezed worlds differ
        pos, vel = seeded_initial_state(spec, g["seeds"][plain])
        assert np.array_equal(pos, g["pos0"][plain])
        assert np.array_equal(vel, g["vel0"][plain])


@pytest.mark.parametrize("name,spec,bench", CASES[:4], ids=[c[0] for c in CASES[:4]])
def test_loop_port_matches_reference(name, spec, bench, golden):
    """The per-object loop (the cpu_baseline port) replays golden worlds one at a time."""
    g = golden(name)
    T, W, A = g["rew"].shape
    env = LoopEnv(spec, benchmark=bench)
    for w in range(min(W, 6)):
        env.set_state(g["pos0"][w], g["vel0"][w])
        for t in range(min(T, 12)):
            obs, rew, done, info = env.step([g["act"][t, w, i] for i in range(A)])
            for i in range(A):
                _close(obs[i], g["obs%d" % i][t, w])
            _close(np.array(rew, dtype=np.float64), g["rew"][t, w])
            assert done == [False] * A
            if name.startswith("simple_spread"):
                assert [x[1] for x in info["n"]] == list(g["info_collisions"][t, w])
                assert [x[3] for x in info["n"]] == list(g["info_occupied"][t, w])
            if name == "simple_tag":
                assert list(info["n"]) == list(g["info_collisions"][t, w])


def test_loop_port_seeded_reset_and_kat():
    """SURVEY.md appendix A.3 known-answer vectors (recorded from the reference, fp64)."""
    env = LoopEnv(ospec.simple_spread(3))
    np.random.seed(0)
    env.reset()
    flat = np.concatenate([p for p in env.pos])
    kat = [0.0976270079, 0.4303787327, 0.2055267521, 0.0897663660, -0.1526904013, 0.2917882261,
           -0.1248255775, 0.7835460016, 0.9273255210, -0.2331169623, 0.5834500762, 0.0577898395]
    assert np.allclose(flat, kat, atol=1e-9)
    act = [np.eye(5)[(i % 4) + 1] for i in range(3)]
    obs, rew, done, _ = env.step(act)
    o0 = [0.6214080569, 0.0672186731, 0.1597678135, 0.4371006001, -0.2845933910, 0.3464454015,
          0.7675577075, -0.6702175624, 0.4236822626, -0.3793107606, -0.0042410614, -0.3473342341,
          -0.3245990205, -0.1020342412, 0, 0, 0, 0]
    assert np.allclose(obs[0], o0, atol=1e-9)
    assert np.allclose(rew, [-8.1422486230] * 3, atol=1e-9)
    for t in range(24):
        act = [np.eye(5)[(i + t) % 5] for i in range(3)]
        obs, rew, done, _ = env.step(act)
    assert np.allclose(env.pos[0], [0.3828512256, 0.5227536118], atol=1e-9)
    assert np.allclose(env.vel[0], [-0.1222033166, 0.4483753008], atol=1e-9)
    assert np.allclose(rew, [-8.4552941772] * 3, atol=1e-9)


def test_contact_kat():
    """Two r=.15 agents at (0,0),(0.2,0), zero action: force (-10,0)/(+10,0)."""
    orc = BatchedOracle(ospec.simple_spread(2, 0), 1)
    orc.set_state([[[0.0, 0.0], [0.2, 0.0]]], np.zeros((1, 2, 2)))
    f = orc.forces(orc.decode(np.zeros((2, 1, 5))))
    assert np.allclose(f[0], [[-10.0, 0.0]], atol=1e-9) and np.allclose(f[1], [[10.0, 0.0]], atol=1e-9)
    orc.integrate(f)
    assert np.allclose(orc.pos[0], [[-0.1, 0.0], [0.3, 0.0]], atol=1e-9)
    assert np.allclose(orc.vel[0], [[-1.0, 0.0], [1.0, 0.0]], atol=1e-9)
    for d, fx in [(0.2999, -0.0744396660), (0.3, -0.0693147181), (0.3001, -0.0644396660)]:
        orc.set_state([[[0.0, 0.0], [d, 0.0]]], np.zeros((1, 2, 2)))
        f = orc.forces(orc.decode(np.zeros((2, 1, 5))))
        assert abs(f[0][0, 0] - fx) < 1e-9


@pytest.mark.parametrize("name,spec,bench", CASES, ids=[c[0] for c in CASES])
def test_c_restatement_matches_reference(name, spec, bench, golden):
    """oracle/mpe_oracle.c (gcc, fp64) replayed from the reference's own states: same arithmetic in the
    same order, so <= 1e-12; collision counts identical."""
    from oracle import build_c
    g = golden(name)
    T, W, A = g["rew"].shape
    for t in range(T):
        p0 = g["pos0"] if t == 0 else g["pos"][t - 1]
        v0 = g["vel0"] if t == 0 else g["vel"][t - 1]
        pos, vel, obs, rew, col = build_c.step_batch(spec, p0, v0, g["act"][t], threads=2)
        _close(pos, g["pos"][t])
        _close(vel, g["vel"][t])
        for i in range(A):
            _close(obs[i], g["obs%d" % i][t])
        _close(rew, g["rew"][t])
        if "info_collisions" in g:
            assert np.array_equal(col, g["info_collisions"][t])


def custom_spec(name, g):
    """The customised constants a custom_*.npz golden was recorded with (tests/golden/gen_golden_custom.py)."""
    import dataclasses
    base = ospec.by_name(name)
    dt, damping, cforce, cmargin = [float(x) for x in g["c_world"]]
    opt = lambda v: None if v < 0 else float(v)
    return dataclasses.replace(base, size=[float(x) for x in g["c_size"]], mass=[float(x) for x in g["c_mass"]],
                               collide=[bool(x) for x in g["c_collide"]], max_speed=[opt(x) for x in g["c_max_speed"]],
                               accel=[opt(x) for x in g["c_accel"]], dt=dt, damping=damping, contact_force=cforce,
                               contact_margin=cmargin)


@pytest.mark.parametrize("name", ["simple_tag", "simple_spread"])
def test_oracles_honour_customised_constants(name, golden):
    """Sizes, masses, collide flags, speed limits, action gains, dt / damping / contact constants changed after
    make_world (plain attributes in the reference, core.py:27-51, 94-99): both restatements against the reference."""
    g = golden("custom_" + name)
    spec = custom_spec(name, g)
    T, W, A = g["rew"].shape
    orc = BatchedOracle(spec, W, np.float64, benchmark=True)
    orc.set_state(g["pos0"], g["vel0"])
    for t in range(T):
        obs, rew, done, info = orc.step(np.transpose(g["act"][t], (1, 0, 2)))
        _close(orc.pos, g["pos"][t])
        _close(orc.vel, g["vel"][t])
        for i in range(A):
            _close(obs[i], g["obs%d" % i][t])
        _close(rew.T, g["rew"][t])
        assert np.array_equal(info["collisions"].T, g["info_collisions"][t])
    from oracle import build_c                       # ... and the plain-C restatement
    for t in range(T):
        p0 = g["pos0"] if t == 0 else g["pos"][t - 1]
        v0 = g["vel0"] if t == 0 else g["vel"][t - 1]
        pos, vel, obs, rew, col = build_c.step_batch(spec, p0, v0, g["act"][t], threads=2)
        _close(pos, g["pos"][t])
        _close(vel, g["vel"][t])
        for i in range(A):
            _close(obs[i], g["obs%d" % i][t])
        _close(rew, g["rew"][t])
        assert np.array_equal(col, g["info_collisions"][t])
    env = LoopEnv(spec, benchmark=True)
    for w in range(4):
        env.set_state(g["pos0"][w], g["vel0"][w])
        for t in range(T):
            obs, rew, done, info = env.step([g["act"][t, w, i] for i in range(A)])
            for i in range(A):
                _close(obs[i], g["obs%d" % i][t, w])
            _close(np.array(rew, dtype=np.float64), g["rew"][t, w])


# ---- the six scenarios outside BASELINE.json's configs: oracle/mpe_f3.py --------------------------------
F3 = ["simple_adversary", "simple_push", "simple_speaker_listener", "simple_reference", "simple_crypto", "simple_world_comm"]


def _replay_f3(spec, g, check_reset=True):
    from oracle.mpe_f3 import F3Oracle
    T, W, A = g["rew"].shape
    assert spec.obs_dims() == [g["obs%d" % i].shape[-1] for i in range(A)]
    orc = F3Oracle(spec, W, np.float64)
    orc.set_state(g["pos0"], g["vel0"])
    orc.set_choice(g["choice"])
    if check_reset:
        for i, o in enumerate(orc.observe()):
            _close(o, g["obs_reset%d" % i])
    worst = 0.0
    for t in range(T):
        obs, rew, done, info = orc.step([g["act%d" % i][t] for i in range(A)])
        _close(orc.pos, g["pos"][t])
        _close(orc.vel, g["vel"][t])
        for i in range(A):
            _close(obs[i], g["obs%d" % i][t])
            _close(orc.c[i], g["c%d" % i][t])
        _close(rew.T, g["rew"][t])
        assert not done.any()
        if "info_collisions" in g:      # simple_world_comm.py:115-124
            assert np.array_equal(info["collisions"].T, g["info_collisions"][t])
        if "info_adv" in g:             # simple_adversary.py:57-67
            _close(info["adv_goal_d2"].T, g["info_adv"][t])
            _close(np.transpose(info["good_d2"], (2, 0, 1)), g["info_good"][t])
    return worst


@pytest.mark.parametrize("name", F3)
def test_f3_oracle_fp64_matches_reference(name, golden):
    """observation / reward / comm state / physics of the six other scenarios, teacher-free replay of the reference's own
    trajectories (the oracle's state follows the reference's to 1e-12 over the whole episode, so every step is compared
    at the reference's states)."""
    _replay_f3(ospec.by_name(name), golden("f3_" + name))


@pytest.mark.parametrize("name", F3)
def test_f3_oracle_honours_customised_constants(name, golden):
    g = golden("f3c_" + name)
    _replay_f3(custom_spec(name, g), g)


@pytest.mark.parametrize("name", F3)
def test_f3_seeded_reset_matches_reference(name, golden):
    """np.random.seed(s); env.reset(): the np.random.choice picks come first, then agents, then landmarks (and
    simple_world_comm's three loops over its landmark groups)."""
    from oracle.mpe_f3 import seeded_initial_state_f3
    g = golden("f3_" + name)
    spec = ospec.by_name(name)
    plain = np.flatnonzero(~g["staged"])      # staged / squeezed worlds were moved after their reset
    pos, vel, choice = seeded_initial_state_f3(spec, g["seeds"])
    assert np.array_equal(choice, g["choice"])
    assert np.array_equal(pos[plain], g["pos0"][plain]) and np.array_equal(vel[plain], g["vel0"][plain])


@pytest.mark.parametrize("name", F3)
def test_f3_goldens_populate_every_branch(name, golden):
    """The discrete branches of the six scenarios' callbacks (goal picks, contacts, speed limits, forests and what they
    hide, the boundary bands, caught prey, food) each hold >= 5 % of the recorded samples: 288 world-steps per scenario
    (round 2) could not say that of simple_world_comm.py:143-289."""
    from oracle.mpe_f3 import branch_coverage
    g = golden("f3_" + name)
    assert g["rew"].shape[1] >= 256
    cov = branch_coverage(ospec.by_name(name), g)
    print("%s coverage: %s" % (name, ", ".join("%s %.1f%%" % (k, 100 * v) for k, v in cov.items())))
    low = {k: v for k, v in cov.items() if v < 0.05}
    assert not low, low


# ---- team sizes other than the reference's make_world (tests/golden/gen_golden_shapes.py) ----------------------------------
SHAPES = ospec.TEAM_SIZE_VARIANTS
shape_spec = ospec.team_size_spec


@pytest.mark.parametrize("name,A,nadv", SHAPES, ids=["%s-%d-%d" % s for s in SHAPES])
def test_f3_oracle_at_other_team_sizes_matches_reference(name, A, nadv, golden):
    """The reference's callbacks on worlds with other team sizes than its make_world builds (its agent / landmark lists
    resized, gen_golden_shapes.py): the oracle replays them at 1e-12, seeded resets included."""
    from oracle.mpe_f3 import seeded_initial_state_f3
    g = golden("shape_%s_%d_%d" % (name, A, nadv))
    spec = shape_spec(name, A, nadv)
    assert (spec.n_agents, sum(spec.adversary)) == (int(g["n_agents"]), int(g["n_adversaries"])) == (A, nadv)
    _replay_f3(spec, g)
    plain = np.flatnonzero(~g["staged"])
    pos, vel, choice = seeded_initial_state_f3(spec, g["seeds"])
    assert np.array_equal(choice, g["choice"])
    assert np.array_equal(pos[plain], g["pos0"][plain]) and np.array_equal(vel[plain], g["vel0"][plain])


# ---- movable landmarks (core.py:158-169 integrates every movable entity) -------------------------------------------------
def movable_spec(name, g):
    import dataclasses
    base = ospec.by_name(name)
    return dataclasses.replace(base, size=[float(x) for x in g["c_size"]], mass=[float(x) for x in g["c_mass"]],
                               collide=[bool(x) for x in g["c_collide"]], movable=[bool(x) for x in g["c_movable"]])


@pytest.mark.parametrize("name", ["simple_tag", "simple_spread"])
def test_oracle_integrates_movable_landmarks_like_the_reference(name, golden):
    """A landmark a user makes movable is pushed by contacts and integrated (its own velocity, damping, mass): the
    batched oracle over the reference's trajectories, all entities' velocities compared."""
    g = golden("movable_" + name)
    spec = movable_spec(name, g)
    T, W, A = g["rew"].shape
    assert g["vel"].shape[2] == spec.n_entities and np.abs(g["vel"][:, :, A:]).max() > 0.1   # the landmark really moves
    orc = BatchedOracle(spec, W, np.float64, benchmark=True)
    orc.set_state(g["pos0"], g["vel0"])
    for t in range(T):
        obs, rew, done, info = orc.step(np.transpose(g["act"][t], (1, 0, 2)))
        _close(orc.pos, g["pos"][t])
        _close(orc.vel, g["vel"][t])
        for i in range(A):
            _close(obs[i], g["obs%d" % i][t])
        _close(rew.T, g["rew"][t])
        assert np.array_equal(info["collisions"].T, g["info_collisions"][t])
