"""GPU parity: the HIP path (through the C ABI / drop-in env) against the oracle and the golden
vectors recorded from the reference.  All tests here need a real MI355X (`-m gpu`).

Bars (BASELINE.json north_star):
  * observations / rewards / positions / velocities: |gpu - fp64 reference| <= 1e-5 * max(1, |ref|)
    PER STEP (teacher-forced: the state is reset to the fp64 trajectory before every step; the
    contact dynamics are chaotic -- error doubles per step in contact -- so free-running
    trajectories are checked against a looser, documented bound).
  * collision counts / occupied landmarks / done flags: bit-exact.  They are strict `<` tests on
    float distances, so "exact" is defined on identical inputs: the counts the GPU emits must
    equal the fp32 oracle's counts evaluated on the positions the GPU emitted (always), and the
    fp64 reference's counts whenever no pair sits within 1e-6 of its threshold (guard band).
"""
import numpy as np
import pytest
import torch

import multiagent_particle_envs_amd as mpe
from oracle import spec as ospec
from oracle.mpe_batched import BatchedOracle, seeded_initial_state

pytestmark = pytest.mark.gpu

TOL = 1e-5


def close(a, b, tol=TOL, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    assert np.all(err <= tol), "%s: max scaled err %.3e at %s" % (what, err.max(), np.unravel_index(err.argmax(), err.shape))
    return float(err.max()) if err.size else 0.0


def np_(t):
    return t.detach().cpu().numpy()


SCN = {
    "simple": (lambda: ospec.simple(), {}),
    "simple_spread": (lambda: ospec.simple_spread(3), {}),
    "simple_tag": (lambda: ospec.simple_tag(), {}),
    "simple_spread_n5": (lambda: ospec.simple_spread(5), {"num_agents": 5}),
    "simple_spread_n64": (lambda: ospec.simple_spread(64), {"num_agents": 64}),
}


def scenario_name(name):
    return "simple_spread" if name.startswith("simple_spread") else name


MASKED = {}   # test -> most worlds the guard band masked (asserted <= 1 % wherever it is used)


def guard_ok(spec, pos64, margin=1e-6, max_frac=0.01):
    """True per world where no counted pair is within `margin` of its collision threshold.  Fails when the band masks
    more than `max_frac` of the worlds (min. one world): a check that masks everything would pass vacuously."""
    ok = _guard_ok(spec, pos64, margin)
    n_masked = int((~ok).sum())
    assert n_masked <= max(1, max_frac * len(ok)), "guard band masks %d of %d worlds" % (n_masked, len(ok))
    import os
    key = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::")[-1]
    MASKED[key] = max(MASKED.get(key, 0), n_masked)
    return ok


def _guard_ok(spec, pos64, margin):
    A = spec.n_agents
    size = np.asarray(spec.size)
    ok = np.ones(pos64.shape[0], bool)

    def near(i_idx, j_idx, thr):
        d = pos64[:, i_idx, None, :] - pos64[:, None, j_idx, :]
        dist = np.sqrt((d ** 2).sum(-1))
        return (np.abs(dist - thr[None]) < margin).any(axis=(1, 2))
    ag = list(range(A))
    if spec.name == "simple_spread":
        ok &= ~near(ag, ag, size[ag][:, None] + size[ag][None, :])
        lm = list(range(A, spec.n_entities))
        ok &= ~near(ag, lm, np.full((A, len(lm)), 0.1))
    if spec.name == "simple_tag":
        good = [j for j in ag if not spec.adversary[j]]
        adv = [j for j in ag if spec.adversary[j]]
        ok &= ~near(good, adv, size[good][:, None] + size[adv][None, :])
    return ok


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(SCN))
def test_teacher_forced_against_reference_golden(name, golden):
    """Every recorded reference step, replayed on the GPU from the reference's own fp64 state."""
    g = golden(name)
    mk, kw = SCN[name]
    spec = mk()
    T, W, A = g["rew"].shape
    env = mpe.make_env(scenario_name(name), benchmark=True, batch_size=W, **kw)
    env.world.set_state(g["pos0"], g["vel0"])
    for t in range(T):
        prev_pos = g["pos0"] if t == 0 else g["pos"][t - 1]
        prev_vel = g["vel0"] if t == 0 else g["vel"][t - 1]
        env.world.set_state(prev_pos, prev_vel)
        act = torch.as_tensor(np.transpose(g["act"][t], (1, 0, 2)), dtype=torch.float32).cuda().contiguous()
        obs_n, rew_n, done_n, info = env.step(act)
        pos, vel = env.world.get_state()
        close(pos, g["pos"][t], what="pos t=%d" % t)
        close(vel, g["vel"][t], what="vel t=%d" % t)
        for i in range(A):
            close(np_(obs_n[i]), g["obs%d" % i][t], what="obs%d t=%d" % (i, t))
            close(np_(rew_n[i]), g["rew"][t][:, i], what="rew%d t=%d" % (i, t))
            assert not np_(done_n[i]).any()
        ok = guard_ok(spec, g["pos"][t])
        if "info_collisions" in g:
            got = np.stack([np_(x[1] if isinstance(x, tuple) else x) for x in info["n"]], axis=1)
            assert np.array_equal(got[ok], g["info_collisions"][t][ok])
        if "info_occupied" in g:
            got = np.stack([np_(x[3]) for x in info["n"]], axis=1)
            assert np.array_equal(got[ok], g["info_occupied"][t][ok])
            close(np.stack([np_(x[2]) for x in info["n"]], axis=1), g["info_min_dists"][t], what="min_dists")
            close(np.stack([np_(x[0]) for x in info["n"]], axis=1), g["info_rew"][t], what="info_rew")


def test_integer_action_ids_against_reference_golden(golden):
    """discrete_action_input=True (environment.py:161-167), incl. its opposite sign convention."""
    g = golden("simple_spread_ids")
    T, W, A = g["rew"].shape
    env = mpe.make_env("simple_spread", benchmark=True, batch_size=W)
    env.discrete_action_input = True
    for t in range(T):
        env.world.set_state(g["pos0"] if t == 0 else g["pos"][t - 1], g["vel0"] if t == 0 else g["vel"][t - 1])
        ids = torch.as_tensor(g["ids"][t].T.copy(), dtype=torch.int32).cuda().contiguous()
        obs_n, rew_n, done_n, info = env.step(ids)
        pos, vel = env.world.get_state()
        close(pos, g["pos"][t])
        for i in range(A):
            close(np_(obs_n[i]), g["obs%d" % i][t])
            close(np_(rew_n[i]), g["rew"][t][:, i])
    # list-of-python-ints form, as a reference caller would pass it
    env.world.set_state(g["pos0"], g["vel0"])
    obs_n, _, _, _ = env.step([g["ids"][0][:, i] for i in range(A)])
    close(np_(obs_n[0]), g["obs0"][0])


# ------------------------------------------------------------------------------------------------
CONFIGS = [
    # (name, oracle spec, scenario kwargs, B, steps) -- BASELINE.json configs[1..3] at full size
    ("simple", lambda: ospec.simple(), {}, 4096, 6),
    ("simple_spread", lambda: ospec.simple_spread(3), {}, 4096, 6),
    ("simple_tag", lambda: ospec.simple_tag(), {}, 16384, 4),
    ("simple_spread", lambda: ospec.simple_spread(6), {"num_agents": 6}, 1024, 3),
    ("simple_spread", lambda: ospec.simple_spread(8), {"num_agents": 8}, 512, 3),       # wide kernel, small N
    ("simple_spread", lambda: ospec.simple_spread(20, 12), {"num_agents": 20, "num_landmarks": 12}, 256, 2),
    ("simple_spread", lambda: ospec.simple_spread(64), {"num_agents": 64}, 96, 2),      # configs[3] arithmetic
    # several worlds per wave (k_multi): 8 / 4 / 2 worlds per wave, ragged last waves, D % 4 != 0 (8-byte pieces)
    ("simple_spread", lambda: ospec.simple_spread(7), {"num_agents": 7}, 1001, 2),
    ("simple_spread", lambda: ospec.simple_spread(16), {"num_agents": 16}, 333, 2),
    ("simple_spread", lambda: ospec.simple_spread(32), {"num_agents": 32}, 131, 2),
    ("simple_spread", lambda: ospec.simple_spread(5, 9), {"num_agents": 5, "num_landmarks": 9}, 257, 2),
    ("simple_spread", lambda: ospec.simple_spread(12, 3), {"num_agents": 12, "num_landmarks": 3}, 100, 2),
    # more than one wave of agents per world (batched contact phase, rows longer than two wave stores), up to the
    # ABI's entity limit MPE_MAX_ENTITIES = 512
    ("simple_spread", lambda: ospec.simple_spread(100), {"num_agents": 100}, 37, 2),
    ("simple_spread", lambda: ospec.simple_spread(70, 3), {"num_agents": 70, "num_landmarks": 3}, 50, 2),
    ("simple_spread", lambda: ospec.simple_spread(3, 90), {"num_agents": 3, "num_landmarks": 90}, 50, 2),
    ("simple_spread", lambda: ospec.simple_spread(256), {"num_agents": 256}, 5, 2),
]


@pytest.mark.parametrize("name,mk,kw,B,steps", CONFIGS, ids=["%s-%s-B%d" % (c[0], "_".join(map(str, c[2].values())) or "ref", c[3]) for c in CONFIGS])
def test_teacher_forced_against_oracle_at_size(name, mk, kw, B, steps, record_parity):
    """Seeded random worlds (a third of them squeezed into contact), random one-hot and soft
    actions; GPU single step from the fp64 state vs the fp64 oracle; integer outputs exact."""
    spec = mk()
    rs = np.random.RandomState(7)
    pos, vel = seeded_initial_state(spec, np.arange(B) + 1000)
    pos[::3] *= 0.35
    vel = rs.uniform(-0.5, 0.5, vel.shape)
    o64 = BatchedOracle(spec, B, np.float64, benchmark=True)
    o32 = BatchedOracle(spec, B, np.float32, benchmark=True)
    o64.set_state(pos, vel)
    env = mpe.make_env(name, benchmark=True, batch_size=B, **kw)
    A = spec.n_agents
    worst = 0.0
    margin = {"pos": 0.0, "vel": 0.0, "obs": 0.0, "rew": 0.0}
    masked = 0
    for t in range(steps):
        # state handed to both sides is the float32-representable rounding of the fp64 trajectory
        p32 = o64.pos.astype(np.float32)
        v32 = o64.vel.astype(np.float32)
        o64.set_state(p32, v32)
        env.world.set_state(p32, v32)
        act = np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(A, B))]
        soft = rs.uniform(-1, 1, size=(A, B, 5)).astype(np.float32)
        act = np.where((rs.rand(A, B) < 0.25)[..., None], soft, act)
        obs64, rew64, done64, info64 = o64.step(act)
        obs_n, rew_n, done_n, info = env.step(torch.as_tensor(act).cuda().contiguous())
        gpos, gvel = env.world.get_state()
        margin["pos"] = max(margin["pos"], close(gpos, o64.pos, what="pos"))
        margin["vel"] = max(margin["vel"], close(gvel, o64.vel, what="vel"))
        for i in range(A):
            margin["obs"] = max(margin["obs"], close(np_(obs_n[i]), obs64[i], what="obs%d" % i))
            margin["rew"] = max(margin["rew"], close(np_(rew_n[i]), rew64[i], what="rew%d" % i))
            assert not np_(done_n[i]).any() and not done64[i].any()
        worst = max(margin.values())
        # integer outputs: exact vs the fp32 oracle on the GPU's own emitted positions ...
        o32.set_state(gpos, gvel)
        _, _, _, info32 = o32.outputs()
        ok = guard_ok(spec, o64.pos)
        masked = max(masked, int((~ok).sum()))
        if spec.name in ("simple_spread", "simple_tag"):
            got = np.stack([np_(x[1] if isinstance(x, tuple) else x) for x in info["n"]], axis=0)
            assert np.array_equal(got, info32["collisions"]), "collision counts differ from fp32 oracle"
            # ... and vs the fp64 reference outside the 1e-6 guard band
            assert np.array_equal(got[:, ok], info64["collisions"][:, ok])
        if spec.name == "simple_spread":
            got = np.stack([np_(x[3]) for x in info["n"]], axis=0)
            assert np.array_equal(got, info32["occupied_landmarks"])
            assert np.array_equal(got[:, ok], info64["occupied_landmarks"][:, ok])
    print("max scaled err %s: %.3e  (%d of %d worlds inside the 1e-6 guard band)" % (name, worst, masked, B))
    assert masked <= max(1, 0.01 * B), "the guard band masks %d of %d worlds: the integer-output check would be vacuous" % (masked, B)
    record_parity("%s_A%d_L%d_B%d" % (name, spec.n_agents, spec.n_landmarks, B),
                  {"worlds": B, "steps": steps, "max_scaled_err": margin, "worlds_masked_guard_band": masked,
                   "integer_outputs": "exact vs fp32 oracle on every world; exact vs fp64 oracle outside the guard band",
                   "against": "oracle/mpe_batched.py fp64, teacher-forced"})


@pytest.mark.parametrize("name,mk,kw,B", [("simple_spread", lambda: ospec.simple_spread(3), {}, 2048),
                                           ("simple_tag", lambda: ospec.simple_tag(), {}, 2048),
                                           ("simple_spread", lambda: ospec.simple_spread(64), {"num_agents": 64}, 512)],
                         ids=["spread3", "tag", "spread64"])
def test_free_running_episode_drift(name, mk, kw, B, record_parity):
    """25 free-running steps (one MADDPG episode), three runs of the same worlds and moves side by side: the fp64 oracle (the
    reference's arithmetic), the SAME arithmetic in float32 (BatchedOracle(np.float32): NumPy's own sqrt / logaddexp / divisions,
    the reference's operation order) and the GPU kernels (regrouped contact force, v_rsq / v_exp / v_log).  fp32 vs fp64 drift
    is amplified by contact stiffness; the question this test answers is whether the kernels' arithmetic drifts
    FASTER than the reference's own would in float32: at t = 5 / 10 / 25 the GPU's drift percentiles (median, p90, p99 over
    worlds) stay within 2x the fp32 oracle's (+ 1e-7: two fp32 ulps of a position)."""
    spec = mk()
    rs = np.random.RandomState(3)
    pos, vel = seeded_initial_state(spec, np.arange(B) + 5000)
    p32 = pos.astype(np.float32)
    o64 = BatchedOracle(spec, B, np.float64)
    o64.set_state(p32, vel)
    o32 = BatchedOracle(spec, B, np.float32)
    o32.set_state(p32, vel)
    env = mpe.make_env(name, batch_size=B, **kw)
    env.world.set_state(p32, vel)
    A = spec.n_agents
    drift, drift32 = {}, {}

    def pcts(err):
        return {"median": float(np.median(err)), "p90": float(np.percentile(err, 90)), "p99": float(np.percentile(err, 99)),
                "max": float(err.max())}
    for t in range(25):
        act = np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(A, B))]
        o64.step(act)
        o32.step(act)
        env.step(torch.as_tensor(act).cuda())
        if t + 1 in (5, 10, 25):
            gpos, gvel = env.world.get_state()
            err = np.abs(gpos - o64.pos).max(axis=(1, 2))
            err32 = np.abs(o32.pos.astype(np.float64) - o64.pos).max(axis=(1, 2))
            drift["t=%d" % (t + 1)], drift32["t=%d" % (t + 1)] = pcts(err), pcts(err32)
    fmt = lambda d: {k: "med %.1e p90 %.1e p99 %.1e max %.1e" % (v["median"], v["p90"], v["p99"], v["max"]) for k, v in d.items()}
    print("%s A=%d free-running vs fp64: GPU %s | NumPy fp32 %s" % (name, A, fmt(drift), fmt(drift32)))
    record_parity("drift_%s_A%d" % (name, A), {"worlds": B, "what": "max |pos - pos_fp64| per world, free-running (no teacher forcing)",
                                               "steps": drift, "steps_numpy_fp32_same_order": drift32})
    for k in drift:
        for q in ("median", "p90", "p99"):
            assert drift[k][q] <= 2.0 * drift32[k][q] + 1e-7, (k, q, drift[k][q], drift32[k][q])
    if A <= 8:
        assert np.median(err) < 1e-5
        assert err.max() < 2e-3
    else:   # N = 64: 64 discs of radius 0.15 in a 2 x 2 box -- every agent is in several stiff contacts at every step, the
        #     episode is chaotic and fp32 / fp64 trajectories separate by ~1e-3 within 25 steps, NumPy's float32 included (the
        #     comparator above is the statement at this size; the per-step teacher-forced bar is what holds at 1e-5)
        assert np.isfinite(err).all() and np.median(err) < 5e-2


# ------------------------------------------------------------------------------------------------
def test_full_size_properties_spread_65536():
    """BASELINE.json's metric configuration (simple_spread N=3, 65536 worlds per GPU): properties
    that do not need the oracle at this size -- determinism, shard invariance (world b's result
    does not depend on which batch it is stepped in: the multi-GPU sharding argument), invariance
    to world order, and the oracle on a strided sample."""
    B = 65536
    spec = ospec.simple_spread(3)
    rs = np.random.RandomState(11)
    pos = rs.uniform(-1, 1, (B, 6, 2)).astype(np.float32)
    pos[::4] *= 0.3
    vel = rs.uniform(-0.3, 0.3, (B, 3, 2)).astype(np.float32)
    act = np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(3, B))]
    env = mpe.make_env("simple_spread", batch_size=B)

    def run(env_, p, v, a):
        env_.world.set_state(p, v)
        o, r, d, _ = env_.step(torch.as_tensor(a).cuda().contiguous())
        ps, vs = env_.world.get_state()
        return [np_(x).copy() for x in o], [np_(x).copy() for x in r], ps, vs
    o1, r1, p1, v1 = run(env, pos, vel, act)
    o2, r2, p2, v2 = run(env, pos, vel, act)
    assert all(np.array_equal(a, b) for a, b in zip(o1, o2)) and np.array_equal(p1, p2)   # deterministic
    # shards: 8 contiguous slices of 8192 (what 8 GPUs would each own) reproduce the full batch bit-for-bit
    env_s = mpe.make_env("simple_spread", batch_size=B // 8)
    for s in range(8):
        sl = slice(s * B // 8, (s + 1) * B // 8)
        os_, rs_, ps_, vs_ = run(env_s, pos[sl], vel[sl], act[:, sl])
        assert all(np.array_equal(os_[i], o1[i][sl]) for i in range(3))
        assert all(np.array_equal(rs_[i], r1[i][sl]) for i in range(3))
        assert np.array_equal(ps_, p1[sl]) and np.array_equal(vs_, v1[sl])
    # permutation of worlds permutes results
    perm = rs.permutation(B)
    o3, r3, p3, v3 = run(env, pos[perm], vel[perm], act[:, perm])
    assert np.array_equal(p3, p1[perm]) and all(np.array_equal(o3[i], o1[i][perm]) for i in range(3))
    # oracle on a strided sample of the full batch
    idx = np.arange(0, B, 37)
    o64 = BatchedOracle(spec, len(idx))
    o64.set_state(pos[idx], vel[idx])
    obs64, rew64, _, _ = o64.step(act[:, idx])
    close(p1[idx], o64.pos)
    for i in range(3):
        close(o1[i][idx], obs64[i])
        close(r1[i][idx], rew64[i])


def test_ragged_batches_and_tiny_batches():
    """B not a multiple of the wave / block / 4 (scalar tail path of the row transpose), B=1."""
    spec = ospec.simple_tag()
    for B in (1, 3, 63, 65, 257, 1000, 1023):
        rs = np.random.RandomState(B)
        pos, vel = seeded_initial_state(spec, np.arange(B) + 77)
        pos[::2] *= 0.3
        o64 = BatchedOracle(spec, B)
        p32 = pos.astype(np.float32)
        o64.set_state(p32, vel)
        env = mpe.make_env("simple_tag", batch_size=B)
        env.world.set_state(p32, vel)
        act = np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(4, B))]
        obs64, rew64, _, _ = o64.step(act)
        obs_n, rew_n, done_n, _ = env.step(torch.as_tensor(act).cuda())
        for i in range(4):
            close(np_(obs_n[i]), obs64[i], what="B=%d obs%d" % (B, i))
            close(np_(rew_n[i]), rew64[i], what="B=%d rew%d" % (B, i))


def test_reset_observation_and_compat_mode(golden):
    """reset() returns the observation of the fresh state; in reference-compatibility mode
    (batch_size=None) `np.random.seed(s); env.reset()` reproduces the reference's own reset."""
    g = golden("simple_spread")
    env = mpe.make_env("simple_spread", benchmark=True)      # compat: B=1, NumPy I/O
    for w in (0, 1, 3, 4):                                     # un-squeezed golden worlds
        np.random.seed(int(g["seeds"][w]))
        obs = env.reset()
        assert isinstance(obs[0], np.ndarray) and obs[0].shape == (18,)
        for i in range(3):
            close(obs[i], g["obs_reset%d" % i][w])
        act = [g["act"][0, w, i] for i in range(3)]
        obs, rew, done, info = env.step(act)
        for i in range(3):
            close(obs[i], g["obs%d" % i][0, w])
        close(np.array(rew), g["rew"][0, w])
        assert done == [False, False, False]
        assert [x[1] for x in info["n"]] == list(g["info_collisions"][0, w])
    # batched reset from per-world NumPy seeds
    envb = mpe.make_env("simple_spread", batch_size=6)
    seeds = [int(g["seeds"][w]) for w in (0, 1, 3, 4, 6, 7)]
    obs = envb.reset(seeds=seeds)
    for i in range(3):
        close(np_(obs[i]), g["obs_reset%d" % i][[0, 1, 3, 4, 6, 7]])


def test_output_lifetime_ping_pong():
    env = mpe.make_env("simple_spread", batch_size=128)
    obs0 = env.reset()
    keep = obs0[0].clone()
    act = torch.zeros((3, 128, 5), device="cuda")
    act[..., 1] = 1.0
    obs1, _, _, _ = env.step(act)
    assert torch.equal(obs0[0], keep)            # previous call's arrays are still intact
    assert not torch.equal(obs1[0], obs0[0])


def test_full_size_properties_spread64_4096():
    """BASELINE.json configs[3] at full size (simple_spread N=64, 4096 worlds): determinism, shard
    invariance of the wave-per-world kernel (4 shards of 1024 reproduce the batch bit-for-bit -- the
    XCD-aware world -> workgroup permutation must not leak into results), the oracle on a sample."""
    B, N = 4096, 64
    spec = ospec.simple_spread(N)
    rs = np.random.RandomState(5)
    pos = rs.uniform(-1, 1, (B, 2 * N, 2)).astype(np.float32)
    pos[::4] *= 0.5
    vel = rs.uniform(-0.3, 0.3, (B, N, 2)).astype(np.float32)
    act = np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(N, B))]
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N, benchmark=True)

    def run(env_, p, v, a):
        env_.world.set_state(p, v)
        o, r, d, info = env_.step(torch.as_tensor(a).cuda().contiguous())
        ps, vs = env_.world.get_state()
        cnt = np.stack([np_(x[1]) for x in info["n"]], axis=0)
        return [np_(x).copy() for x in o], [np_(x).copy() for x in r], ps, vs, cnt
    o1, r1, p1, v1, c1 = run(env, pos, vel, act)
    o2, r2, p2, v2, c2 = run(env, pos, vel, act)
    assert np.array_equal(p1, p2) and np.array_equal(c1, c2) and all(np.array_equal(a, b) for a, b in zip(o1, o2))
    env_s = mpe.make_env("simple_spread", batch_size=B // 4, num_agents=N, benchmark=True)
    for s_ in range(4):
        sl = slice(s_ * B // 4, (s_ + 1) * B // 4)
        os_, rs_, ps_, vs_, cs_ = run(env_s, pos[sl], vel[sl], act[:, sl])
        assert np.array_equal(ps_, p1[sl]) and np.array_equal(vs_, v1[sl]) and np.array_equal(cs_, c1[:, sl])
        for i in (0, 17, 63):
            assert np.array_equal(os_[i], o1[i][sl]) and np.array_equal(rs_[i], r1[i][sl])
    idx = np.arange(3, B, 8)             # 512 worlds, every residue of the world -> workgroup map (4 worlds per workgroup, XCD-aware order)
    o64 = BatchedOracle(spec, len(idx), benchmark=True)
    o64.set_state(pos[idx], vel[idx])
    obs64, rew64, _, info64 = o64.step(act[:, idx])
    close(p1[idx], o64.pos)
    close(v1[idx], o64.vel)
    for i in range(N):
        close(o1[i][idx], obs64[i])
        close(r1[i][idx], rew64[i])
    ok = guard_ok(spec, o64.pos)
    assert ok.sum() >= 0.9 * len(idx)
    assert np.array_equal(c1[:, idx][:, ok], info64["collisions"][:, ok])


def test_spread64_reference_worlds_inside_the_real_grid(golden, record_parity):
    """tests/golden/simple_spread_n64_w64.npz: 64 worlds x 3 steps recorded from the REFERENCE at N = 64, stepped here as 64 of
    the 4096 worlds of BASELINE's C4 env (the others are seeded filler that keeps evolving) -- the two-waves-per-world kernel at
    its real grid, its XCD-aware world -> workgroup map and cooperative staging, against reference data: teacher-forced, state /
    rewards / rows of agents 0, 17, 63 at 1e-5, contact counts exact outside the guard band.

    The kernel is handed the float32 ROUNDING of the reference's float64 state while the reference stepped the unrounded one, so
    the distance to the reference mixes two things (round-5 verdict #3; tools/c4_parity_ab.py, profiles/r6_c4_parity_ab.txt):
    the input rounding (6e-8 relative on a position x contact stiffness <= 100 x dt x the overlapping partners: 8.9e-6 on this
    fixture, at a velocity of 0.0034 with many partners -- ANY float32 evaluation pays it, NumPy's included) and the kernel's own
    arithmetic.  The control separates them: the fp64 oracle stepped from the SAME rounded state is the kernel's error alone,
    held to 3e-6 here (measured 1.8e-6; the reference's operation order in NumPy float32: 1.4e-6)."""
    g = golden("simple_spread_n64_w64")
    T, W, N = g["rew"].shape
    B = 4096
    spec = ospec.simple_spread(N)
    slots = (np.arange(W) * 61 + 5) % B                  # scattered over the batch: 61 is odd, every residue mod 4 and mod 8 occurs
    assert len(set(slots.tolist())) == W
    rs = np.random.RandomState(8)
    pos = rs.uniform(-1, 1, (B, 2 * N, 2))
    vel = np.zeros((B, N, 2))
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N, benchmark=True)
    worst, worst_at, kernel_only, rounding_only = 0.0, None, 0.0, 0.0

    def where(a, b, what):
        e = np.abs(np.asarray(a, np.float64) - b) / np.maximum(1.0, np.abs(b))
        k = np.unravel_index(int(e.argmax()), e.shape)
        return float(e.max()), "%s at (world, agent, xy) %s, reference value %+.4f" % (what, tuple(int(x) for x in k), float(b[k]))
    for t in range(T):
        p0, v0 = (g["pos0"], g["vel0"]) if t == 0 else (g["pos"][t - 1], g["vel"][t - 1])
        pos[slots], vel[slots] = p0, v0
        env.world.set_state(pos, vel)
        act = np.eye(5)[rs.randint(0, 5, size=(N, B))]
        act[:, slots] = np.transpose(g["act"][t], (1, 0, 2))
        obs_n, rew_n, _, info = env.step(torch.as_tensor(act, dtype=torch.float32).cuda().contiguous())
        pos, vel = env.world.get_state()
        pos, vel = pos.astype(np.float64), vel.astype(np.float64)
        close(pos[slots], g["pos"][t], what="pos t=%d" % t)
        close(vel[slots], g["vel"][t], what="vel t=%d" % t)
        for e, at in (where(pos[slots], g["pos"][t], "pos t=%d" % t), where(vel[slots], g["vel"][t], "vel t=%d" % t)):
            if e > worst:
                worst, worst_at = e, at
        for i in (int(a) for a in g["obs_agents"]):
            worst = max(worst, close(np_(obs_n[i])[slots], g["obs%d" % i][t], what="obs%d t=%d" % (i, t)))
        ok = guard_ok(spec, g["pos"][t])
        cnt = np.stack([np_(x[1]) for x in info["n"]], axis=1)[slots]       # [W, N]
        assert np.array_equal(cnt[ok], g["info_collisions"][t][ok])
        for i in range(N):
            worst = max(worst, close(np_(rew_n[i])[slots][ok], g["rew"][t][:, i][ok], what="rew%d t=%d" % (i, t)))
        # the control: the fp64 oracle from the float32-ROUNDED state -- what the kernel was handed
        o = BatchedOracle(spec, W)
        o.set_state(p0.astype(np.float32).astype(np.float64), v0.astype(np.float32).astype(np.float64))
        _, rew_o, _, _ = o.step(act[:, slots])
        kernel_only = max(kernel_only, close(pos[slots], o.pos, tol=3e-6, what="pos vs oracle from the rounded state t=%d" % t),
                          close(vel[slots], o.vel[:, :N], tol=3e-6, what="vel vs oracle from the rounded state t=%d" % t),
                          close(np.stack([np_(r) for r in rew_n], axis=1)[slots], np.stack(rew_o, axis=1), tol=3e-6, what="rew vs oracle t=%d" % t))
        rounding_only = max(rounding_only, where(o.pos, g["pos"][t], "")[0], where(o.vel[:, :N], g["vel"][t], "")[0])
    assert kernel_only < 0.5 * worst or worst < 3e-6        # (the margin goes to the input rounding, not to the kernel)
    record_parity("simple_spread_n64_reference_worlds_in_B4096", {
        "worlds": W, "of": B, "steps": T, "max_scaled_err": worst, "max_scaled_err_where": worst_at,
        "kernel_only_max_scaled_err": kernel_only, "input_rounding_alone_max_scaled_err": rounding_only,
        "what": "max_scaled_err: kernel (fed the float32 rounding of the reference's float64 state) vs the reference; kernel_only: vs "
                "the fp64 oracle stepped from that SAME rounded state; input_rounding_alone: that oracle vs the reference",
        "against": "tests/golden/simple_spread_n64_w64.npz (the reference itself), teacher-forced"})


def test_a_degenerate_world_does_not_poison_its_neighbours():
    """Two agents at the same point give d = 0 -> NaN forces in the reference (core.py:193).
    The NaNs must stay inside that world: its neighbours in the same wave / workgroup are unaffected."""
    for name, kw, n_agents in (("simple_spread", {}, 3), ("simple_spread", {"num_agents": 16}, 16)):
        B = 256
        spec = ospec.simple_spread(n_agents)
        pos, vel = seeded_initial_state(spec, np.arange(B) + 31)
        p32 = pos.astype(np.float32)
        bad = [5, 64, 200]
        for w in bad:
            p32[w, 1] = p32[w, 0]                   # agent 1 on top of agent 0
        act = np.eye(5, dtype=np.float32)[np.random.RandomState(2).randint(0, 5, size=(n_agents, B))]
        env = mpe.make_env(name, batch_size=B, **kw)
        env.world.set_state(p32, vel)
        obs_n, rew_n, _, _ = env.step(torch.as_tensor(act).cuda())
        gpos, _ = env.world.get_state()
        good = np.ones(B, bool)
        good[bad] = False
        assert np.isnan(gpos[bad]).any(axis=(1, 2)).all()          # the reference's NaN is reproduced ...
        assert np.isfinite(gpos[good]).all()                       # ... and contained
        o64 = BatchedOracle(spec, int(good.sum()))
        o64.set_state(p32[good], vel[good])
        obs64, rew64, _, _ = o64.step(act[:, good])
        close(gpos[good], o64.pos)
        for i in range(n_agents):
            close(np_(obs_n[i])[good], obs64[i])
            close(np_(rew_n[i])[good], rew64[i])


def test_batch_multi_agent_env_concatenates_per_agent_lists():
    """environment.py:288-335: lists of several envs concatenated; the reference's stray `time` argument is accepted."""
    e1 = mpe.make_env("simple_spread", batch_size=8)
    e2 = mpe.make_env("simple_tag", batch_size=8)
    both = mpe.BatchMultiAgentEnv([e1, e2])
    assert both.n == 7
    obs = both.reset()
    assert [o.shape[1] for o in obs] == [18, 18, 18, 16, 16, 16, 14]
    act = [torch.zeros((8, 5), device="cuda") for _ in range(7)]
    obs, rew, done, info = both.step(act, 0)
    assert len(obs) == len(rew) == len(done) == 7 and info == {"n": []}


@pytest.mark.parametrize("N,B", [(16, 300), (40, 90)])
def test_large_worlds_other_entry_paths(N, B):
    """The large-N kernels (several worlds per wave for N=16, a wave per world for N=40) through their other
    doors: integer action ids (discrete_action_input; opposite sign convention), and World.step alone
    (mpe_world_step) under a user scenario whose callbacks stay in Python -- against the fused step."""
    rs = np.random.RandomState(N)
    pos = rs.uniform(-1, 1, (B, 2 * N, 2)).astype(np.float32)
    pos[::2] *= 0.5
    vel = rs.uniform(-0.4, 0.4, (B, N, 2)).astype(np.float32)
    ids = rs.randint(0, 5, size=(N, B))
    swap = np.array([0, 2, 1, 4, 3])                       # id 1 = -x, one-hot index 1 = +x
    rows = torch.as_tensor(np.eye(5, dtype=np.float32)[swap[ids]]).cuda()
    ref = mpe.make_env("simple_spread", batch_size=B, num_agents=N)
    ref.world.set_state(pos, vel)
    o_ref, r_ref, _, _ = ref.step(rows)
    p_ref, v_ref = ref.world.get_state()
    # integer ids
    e_id = mpe.make_env("simple_spread", batch_size=B, num_agents=N)
    e_id.discrete_action_input = True
    e_id.world.set_state(pos, vel)
    o_id, r_id, _, _ = e_id.step(torch.as_tensor(ids.astype(np.int32)).cuda())
    p_id, v_id = e_id.world.get_state()
    assert np.array_equal(p_id, p_ref) and np.array_equal(v_id, v_ref)
    for i in (0, N // 2, N - 1):
        assert torch.equal(o_id[i], o_ref[i]) and torch.equal(r_id[i], r_ref[i])
    # generic path: the same scenario class with an overridden (equivalent) reward -> torch callbacks + mpe_world_step
    Base = mpe.scenarios.load("simple_spread.py").Scenario

    class Mine(Base):
        def reward(self, agent, world):
            return Base.reward(self, agent, world)
    sc = Mine()
    w = sc.make_world(batch_size=B, num_agents=N)
    e_gen = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, fused=False)
    assert not e_gen.fused
    w.set_state(pos, vel)
    o_g, r_g, _, _ = e_gen.step([rows[i] for i in range(N)])
    p_g, v_g = w.get_state()
    assert np.array_equal(p_g, p_ref) and np.array_equal(v_g, v_ref)     # the same physics kernel arithmetic
    for i in (0, N - 1):
        close(np_(o_g[i]), np_(o_ref[i]))
        close(np_(r_g[i]), np_(r_ref[i]))


def test_large_all_colliding_world_against_oracle():
    """A 90-entity simple_tag (40 predators, 30 prey, 20 obstacles): every entity collides, three different sizes,
    speed clamps -- the wave-per-world physics with a ranked partner list that is NOT the agent block, more than
    one wave of agents, non-uniform reach.  Whatever path the env takes for this shape must match the oracle."""
    B = 64
    spec = ospec.simple_tag(40, 30, 20)
    env = mpe.make_env("simple_tag", batch_size=B, num_adversaries=40, num_good_agents=30, num_landmarks=20)
    A = spec.n_agents
    rs = np.random.RandomState(2)
    pos = rs.uniform(-1, 1, (B, spec.n_entities, 2))
    pos[::2] *= 0.4                                            # crowded worlds: many simultaneous contacts
    vel = rs.uniform(-1.2, 1.2, (B, A, 2))
    o64 = BatchedOracle(spec, B)
    p32, v32 = pos.astype(np.float32), vel.astype(np.float32)
    o64.set_state(p32, v32)
    env.world.set_state(p32, v32)
    act = np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(A, B))]
    obs64, rew64, _, _ = o64.step(act)
    obs_n, rew_n, done_n, _ = env.step(torch.as_tensor(act).cuda())
    gp, gv = env.world.get_state()
    close(gp, o64.pos, what="pos")          # the per-element bar of every other parity test: 1e-5 * max(1, |ref|)
    close(gv, o64.vel, what="vel")
    ok = guard_ok(spec, o64.pos)
    for i in range(A):
        close(np_(obs_n[i]), obs64[i], what="obs%d" % i)
        close(np_(rew_n[i])[ok], rew64[i][ok], what="rew%d" % i)


def test_world_step_at_the_entity_limit():
    """MPE_MAX_ENTITIES = 512 entities, all colliding, three sizes, speed clamps (a 200 + 112 + 200 simple_tag):
    `World.step` alone (mpe_world_step -> the wave-per-world kernel with > 64 KiB of LDS per workgroup)."""
    B = 6
    spec = ospec.simple_tag(200, 112, 200)
    sc = mpe.scenarios.load("simple_tag.py").Scenario()
    w = sc.make_world(batch_size=B, num_adversaries=200, num_good_agents=112, num_landmarks=200)
    A = spec.n_agents
    rs = np.random.RandomState(3)
    pos = rs.uniform(-1, 1, (B, spec.n_entities, 2)).astype(np.float32)
    vel = rs.uniform(-1.2, 1.2, (B, A, 2)).astype(np.float32)
    o64 = BatchedOracle(spec, B)
    o64.set_state(pos, vel)
    w.set_state(pos, vel)
    act = np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(A, B))]
    o64.integrate(o64.forces(o64.decode(act)))            # World.step of the oracle (no outputs)
    for i, agent in enumerate(w.agents):
        a = torch.as_tensor(act[i]).cuda()
        agent.action.u = torch.stack([a[:, 1] - a[:, 2], a[:, 3] - a[:, 4]], dim=1) * agent.accel
    w.step()
    gp, gv = w.get_state()
    close(gp, o64.pos, what="pos")
    close(gv, o64.vel, what="vel")


@pytest.mark.parametrize("name", ["simple_tag", "simple_spread"])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "generic"])
def test_customised_constants_against_reference_golden(name, fused, golden):
    """tests/golden/custom_*.npz: the reference stepped after its entity / world attributes were changed (sizes, masses,
    collide flags, speed limits, action gains, dt, damping, contact force and margin: core.py:27-51, 94-99).  The same
    attribute assignments on this package's world, teacher-forced from the reference's fp64 states -- through the fused
    kernel (run-time descriptor constants, `if agent.collide` reward gates) and through the generic path."""
    from test_oracle_golden import custom_spec
    g = golden("custom_" + name)
    spec = custom_spec(name, g)
    T, W, A = g["rew"].shape
    sc = mpe.scenarios.load(name + ".py").Scenario()
    w = sc.make_world(batch_size=W)
    for k, e in enumerate(w.entities):
        e.size, e.initial_mass, e.collide = float(g["c_size"][k]), float(g["c_mass"][k]), bool(g["c_collide"][k])
    for k, a in enumerate(w.agents):
        a.max_speed = None if g["c_max_speed"][k] < 0 else float(g["c_max_speed"][k])
        a.accel = None if g["c_accel"][k] < 0 else float(g["c_accel"][k])
    w.dt, w.damping, w.contact_force, w.contact_margin = [float(x) for x in g["c_world"]]
    env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, sc.benchmark_data, fused=fused)
    for t in range(T):
        prev_pos = g["pos0"] if t == 0 else g["pos"][t - 1]
        prev_vel = g["vel0"] if t == 0 else g["vel"][t - 1]
        w.set_state(prev_pos, prev_vel)
        act = torch.as_tensor(np.transpose(g["act"][t], (1, 0, 2)), dtype=torch.float32).cuda().contiguous()
        obs_n, rew_n, _, info = env.step(act if fused else [act[i] for i in range(A)])
        pos, vel = w.get_state()
        close(pos, g["pos"][t], what="pos t=%d" % t)
        close(vel, g["vel"][t], what="vel t=%d" % t)
        ok = guard_ok(spec, g["pos"][t])
        for i in range(A):
            close(np_(obs_n[i]), g["obs%d" % i][t], what="obs%d t=%d" % (i, t))
            close((np_(rew_n[i]) * np.ones(W))[ok], g["rew"][t][:, i][ok], what="rew%d t=%d" % (i, t))
        cols = [x[1] if isinstance(x, tuple) else x for x in info["n"]]       # a plain 0 where `agent.collide` is off
        got = np.stack([np.asarray(np_(c) if torch.is_tensor(c) else c) * np.ones(W, np.int64) for c in cols], axis=1)
        assert np.array_equal(got[ok], g["info_collisions"][t][ok])


@pytest.mark.parametrize("name", ["simple_tag", "simple_spread"])
@pytest.mark.parametrize("when", ["before", "live"])
def test_movable_landmark_against_reference_golden(name, when, golden, record_parity):
    """tests/golden/movable_*.npz: the reference with a landmark made movable (simple_tag's obstacle 0 a ball of mass 3;
    simple_spread's landmark 1 a colliding ball of mass 0.5 between two immovable ones).  core.py:158-169 integrates every
    movable entity and core.py:194-195 pushes both sides of a contact: the landmark's position AND velocity are compared
    at 1e-5, teacher-forced, next to the agents'.  `mpe_world_step` steps the entities up to the last movable one as
    action-less agents (same entity order); the scenario's callbacks run on the post-step world.  'live': the landmark
    is made movable on a fused env that is already stepping -- the env must leave the fused path by itself."""
    from test_oracle_golden import movable_spec
    g = golden("movable_" + name)
    spec = movable_spec(name, g)
    T, W, A = g["rew"].shape
    E = spec.n_entities

    def customise(w):
        for k, e in enumerate(w.entities):
            e.size, e.initial_mass, e.collide, e.movable = float(g["c_size"][k]), float(g["c_mass"][k]), bool(g["c_collide"][k]), bool(g["c_movable"][k])
    if when == "before":
        sc = mpe.scenarios.load(name + ".py").Scenario()
        w = sc.make_world(batch_size=W)
        customise(w)
        env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, sc.benchmark_data)
    else:
        env = mpe.make_env(name, batch_size=W, benchmark=True)
        w = env.world
        assert env.fused
        env.reset()
        env.step(torch.zeros((A, W, 5), device="cuda"))
        customise(w)
    worst = 0.0
    for t in range(T):
        w.set_state(g["pos0"] if t == 0 else g["pos"][t - 1], g["vel0"] if t == 0 else g["vel"][t - 1])
        act = torch.as_tensor(np.transpose(g["act"][t], (1, 0, 2)), dtype=torch.float32).cuda().contiguous()
        obs_n, rew_n, _, info = env.step([act[i] for i in range(A)])
        assert not env.fused
        pos, vel = w.get_state(all_entities=True)
        assert vel.shape == (W, E, 2)
        worst = max(worst, close(pos, g["pos"][t], what="pos t=%d" % t), close(vel, g["vel"][t], what="vel t=%d" % t))
        ok = guard_ok(spec, g["pos"][t])
        for i in range(A):
            worst = max(worst, close(np_(obs_n[i]), g["obs%d" % i][t], what="obs%d t=%d" % (i, t)))
            worst = max(worst, close((np_(rew_n[i]) * np.ones(W))[ok], g["rew"][t][:, i][ok], what="rew%d t=%d" % (i, t)))
    lm = [k for k in range(A, E) if g["c_movable"][k]]
    assert np.abs(vel[:, lm]).max() > 0.05          # it moved here too
    record_parity("movable_landmark_%s_%s" % (name, when), {"worlds": W, "steps": T, "max_scaled_err": worst,
                                                             "against": "tests/golden/movable_%s.npz (the reference itself), teacher-forced" % name})


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "generic"])
def test_force_discrete_and_continuous_action_modes_against_reference_golden(fused, golden):
    """environment.py:169-172 (force_discrete_action: the soft row becomes the one-hot of its argmax) and :176-177
    (discrete_action_space = False: the action IS the force direction, two numbers) -- recorded from the reference
    (tests/golden/gen_golden_custom.py --modes), teacher-forced here."""
    g = golden("mode_force_discrete")
    T, W, A = g["rew"].shape
    env = mpe.make_env("simple_spread", batch_size=W, fused=fused)
    env.force_discrete_action = True
    for t in range(T):
        env.world.set_state(g["pos0"] if t == 0 else g["pos"][t - 1], g["vel0"] if t == 0 else g["vel"][t - 1])
        act = torch.as_tensor(np.transpose(g["act"][t], (1, 0, 2)), dtype=torch.float32).cuda().contiguous()
        keep = act.clone()
        obs_n, rew_n, _, _ = env.step(act if fused else [act[i] for i in range(A)])
        pos, vel = env.world.get_state()
        close(pos, g["pos"][t], what="pos t=%d" % t)
        close(vel, g["vel"][t], what="vel t=%d" % t)
        for i in range(A):
            close(np_(obs_n[i]), g["obs%d" % i][t], what="obs%d" % i)
    g = golden("mode_continuous")
    T, W, A = g["rew"].shape
    env = mpe.make_env("simple_tag", batch_size=W, fused=fused)
    env.discrete_action_space = False
    for t in range(T):
        env.world.set_state(g["pos0"] if t == 0 else g["pos"][t - 1], g["vel0"] if t == 0 else g["vel"][t - 1])
        act = [torch.as_tensor(g["act2"][t][:, i], dtype=torch.float32).cuda() for i in range(A)]
        obs_n, rew_n, _, _ = env.step(act)
        pos, vel = env.world.get_state()
        close(pos, g["pos"][t], what="continuous pos t=%d" % t)
        close(vel, g["vel"][t], what="continuous vel t=%d" % t)
        for i in range(A):
            close(np_(obs_n[i]), g["obs%d" % i][t], what="continuous obs%d" % i)


@pytest.mark.parametrize("teams,B", [((5, 2, 1), 2048), ((2, 2, 3), 1000), ((40, 30, 20), 64), ((70, 10, 5), 33),
                                     ((1, 6, 1), 500)])
def test_simple_tag_any_team_sizes_fused_against_oracle(teams, B):
    """simple_tag.py:84-147 are N-generic; only make_world hard-codes 3 / 1 / 2.  Every other split runs the
    wave-per-world kernel's tag stage (ragged rows, team rewards, benchmark counts): per-element bar vs the oracle."""
    nadv, ngood, nl = teams
    spec = ospec.simple_tag(nadv, ngood, nl)
    env = mpe.make_env("simple_tag", benchmark=True, batch_size=B, num_adversaries=nadv, num_good_agents=ngood,
                       num_landmarks=nl)
    assert env.fused
    A = spec.n_agents
    rs = np.random.RandomState(nadv * 100 + ngood)
    pos = rs.uniform(-1.1, 1.1, (B, spec.n_entities, 2))      # some prey beyond the +-0.9 / +-1 boundary penalties
    pos[::3] *= 0.35
    vel = rs.uniform(-1.2, 1.2, (B, A, 2))
    p32, v32 = pos.astype(np.float32), vel.astype(np.float32)
    o64 = BatchedOracle(spec, B, benchmark=True)
    o64.set_state(p32, v32)
    env.world.set_state(p32, v32)
    for t in range(2):
        act = np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(A, B))]
        st_p, st_v = o64.pos.copy(), o64.vel.copy()
        env.world.set_state(st_p.astype(np.float32), st_v.astype(np.float32))     # teacher-forced
        o64.set_state(st_p.astype(np.float32), st_v.astype(np.float32))
        obs64, rew64, _, info64 = o64.step(act)
        obs_n, rew_n, done_n, info = env.step(torch.as_tensor(act).cuda())
        gp, gv = env.world.get_state()
        close(gp, o64.pos, what="pos")
        close(gv, o64.vel, what="vel")
        ok = guard_ok(spec, o64.pos)
        for i in range(A):
            assert obs_n[i].shape == (B, obs64[i].shape[1])
            close(np_(obs_n[i]), obs64[i], what="obs%d" % i)
            close(np_(rew_n[i])[ok], rew64[i][ok], what="rew%d" % i)
            assert not np_(done_n[i]).any()
        got = np.stack([np_(x) for x in info["n"]], axis=1)          # [B, A] benchmark_data collision counts
        assert np.array_equal(got[ok], info64["collisions"].T[ok])


def test_simple_tag_fused_equals_generic_at_65536_worlds():
    """The wave-per-world tag stage against the scenario's own torch callbacks (generic path: same physics kernel)."""
    B, kw = 65536, dict(num_adversaries=5, num_good_agents=2, num_landmarks=1)
    ef = mpe.make_env("simple_tag", batch_size=B, seed=3, **kw)
    eg = mpe.make_env("simple_tag", batch_size=B, seed=3, fused=False, **kw)
    assert ef.fused and not eg.fused
    eg.world.pos.copy_(ef.world.pos)
    eg.world.vel.copy_(ef.world.vel)
    g = torch.Generator(device="cuda").manual_seed(1)
    for t in range(3):
        act = torch.nn.functional.one_hot(torch.randint(0, 5, (7, B), device="cuda", generator=g), 5).float()
        of, rf, _, _ = ef.step(act)
        og, rg, _, _ = eg.step([act[i] for i in range(7)])
        assert torch.equal(ef.world.pos, eg.world.pos) and torch.equal(ef.world.vel, eg.world.vel)
        for i in range(7):
            close(np_(of[i]), np_(og[i]), what="obs%d" % i)
            close(np_(rf[i]), np_(rg[i]), what="rew%d" % i)


def test_two_waves_per_world_kernel_is_bit_identical_to_the_wave_per_world_kernel():
    """`mpe_step` on 33..64 identical agents runs k_duo (one workgroup of two waves per world); `mpe_world_step` followed
    by `mpe_observe` runs k_wave's physics and output stages on the same state.  Same arithmetic, same order: equal bits.
    B is not a multiple of the 256-world blocks of k_duo's XCD map; N = 40 has dead lanes, N = 64 none."""
    import ctypes as C
    from multiagent_particle_envs_amd import _abi
    L_ = _abi.lib()
    for N, B in ((64, 300), (40, 77), (33, 1000), (63, 41), (64, 1), (48, 5), (64, 4096)):   # odd N: 8-byte row pieces; B < 4: ragged groups; (64, 4096) = BASELINE config C4 at full size
        rs = np.random.RandomState(N)
        pos = rs.uniform(-1, 1, (B, 2 * N, 2)).astype(np.float32)
        pos[::2] *= 0.5
        vel = rs.uniform(-1, 1, (B, N, 2)).astype(np.float32)
        act = torch.as_tensor(rs.uniform(-1, 1, (N, B, 5)).astype(np.float32)).cuda()
        e1 = mpe.make_env("simple_spread", benchmark=True, batch_size=B, num_agents=N)
        e2 = mpe.make_env("simple_spread", benchmark=True, batch_size=B, num_agents=N)
        for e in (e1, e2):
            e.world.set_state(pos, vel)
            e._ensure_buffers()
        o1, r1, _, i1 = e1.step(act)
        out = e2._sets[0]
        b = out.bufs
        b.act, b.ids, b.u = act.data_ptr(), None, None
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _abi.check(L_.mpe_world_step(C.byref(e2._desc), C.byref(b), B, st), "mpe_world_step")
        b.act = None
        _abi.check(L_.mpe_observe(C.byref(e2._desc), C.byref(b), B, st), "mpe_observe")
        assert torch.equal(e1.world.pos, e2.world.pos) and torch.equal(e1.world.vel, e2.world.vel)
        for i in range(N):
            assert torch.equal(o1[i], out.obs_n[i]), (N, i)
            assert torch.equal(r1[i], out.reward_n[i]), (N, i)
        for k in ("rew", "collisions", "min_dists", "occupied_landmarks"):
            assert torch.equal(e1._sets[e1._flip].info[k], out.info[k]), k


@pytest.mark.parametrize("name,kw,bench", [("simple_spread", {}, False), ("simple_tag", {}, True), ("simple_spread", {"num_agents": 20}, False)])
def test_step_fast_path_is_the_same_step(name, kw, bench):
    """MultiAgentEnv.step has a short path for the caller who passes the SAME preallocated [A,B,5] device tensor step after
    step (contents rewritten in place).  It must be the very same step: bit-identical to an env that is handed a fresh
    tensor every time, and it must stand down whenever something was flipped on the env or the world in between."""
    B = 513
    fast = mpe.make_env(name, batch_size=B, seed=11, benchmark=bench, **kw)
    slow = mpe.make_env(name, batch_size=B, seed=11, benchmark=bench, **kw)
    A = fast.n
    act = torch.zeros((A, B, 5), device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)

    def both(t):
        act.copy_(torch.rand((A, B, 5), device="cuda", generator=g))
        of, rf, df, nf = fast.step(act)
        os_, rs, ds, ns = slow.step(act.clone())
        assert torch.equal(fast.world.pos, slow.world.pos) and torch.equal(fast.world.vel, slow.world.vel), t
        for i in range(A):
            assert torch.equal(of[i], os_[i]) and torch.equal(rf[i], rs[i]) and torch.equal(df[i], ds[i]), (t, i)
            a, b = nf["n"][i], ns["n"][i]
            for x, y in zip(a if isinstance(a, tuple) else (a,), b if isinstance(b, tuple) else (b,)):
                assert (not torch.is_tensor(x)) or torch.equal(x, y)
    for t in range(4):
        both(t)
        assert id(act) in fast._fast_acts and id(act) not in slow._fast_acts
    for e in (fast, slow):                         # a constant assigned on a live env: honoured at the next step
        e.world.agents[0].size = 0.4
    both("after a constant changed")
    both("and again")
    for e in (fast, slow):
        e.force_discrete_action = True             # environment.py:169-172: needs the argmax pass -> not the short path
    both("force_discrete_action")
    for e in (fast, slow):
        e.force_discrete_action = False
    o1, o2 = fast.reset(seeds=list(range(B))), slow.reset(seeds=list(range(B)))
    for i in range(A):
        assert torch.equal(o1[i], o2[i])
    both("after a reset")
    both("armed after the reset")
    ring = [torch.rand((A, B, 5), device="cuda", generator=g) for _ in range(3)]     # a caller cycling through a few buffers
    for t in range(7):
        of = fast.step(ring[t % 3])[0]
        os_ = slow.step(ring[t % 3].clone())[0]
        assert all(torch.equal(a, b) for a, b in zip(of, os_)), t
    assert all(id(r) in fast._fast_acts for r in ring)
    k = id(ring[0])
    del ring, of
    import gc
    gc.collect()
    assert fast._fast_acts[k]() is None            # remembered by weak reference only
    # outputs are still the ping-pong views: the rows of step t stay valid through step t + 1
    act.copy_(torch.rand((A, B, 5), device="cuda", generator=g))
    o_t = fast.step(act)[0]
    keep = [x.clone() for x in o_t]
    fast.step(act)
    assert all(torch.equal(a, b) for a, b in zip(o_t, keep))


def test_a_list_of_the_rows_of_one_tensor_is_used_in_place():
    """The reference's per-agent list, when it is `[t[0], t[1], ...]` of one [A,B,5] device tensor (a stacked policy output):
    no staging copies, the short path of the tensor form -- and the very same step as independent per-agent tensors."""
    B = 640
    a, b = mpe.make_env("simple_tag", batch_size=B, seed=3), mpe.make_env("simple_tag", batch_size=B, seed=3)
    A = a.n
    t = torch.zeros((A, B, 5), device="cuda")
    g = torch.Generator(device="cuda").manual_seed(2)
    for k in range(5):
        t.copy_(torch.rand((A, B, 5), device="cuda", generator=g))
        oa, ra, _, _ = a.step([t[i] for i in range(A)])                 # rows of one tensor
        ob, rb, _, _ = b.step([t[i].clone() for i in range(A)])         # independent tensors
        assert torch.equal(a.world.pos, b.world.pos) and all(torch.equal(x, y) for x, y in zip(oa + ra, ob + rb)), k
    assert id(t) in a._fast_acts and not b._fast_acts
    # not the rows of one tensor (another order; a slice of something larger): staged as before, same results
    big = torch.rand((A + 1, B, 5), device="cuda", generator=g)
    for rows in ([t[i] for i in reversed(range(A))], [big[i] for i in range(A)]):
        oa = a.step(rows)[0]
        ob = b.step([r.clone() for r in rows])[0]
        assert all(torch.equal(x, y) for x, y in zip(oa, ob))


@pytest.mark.parametrize("ids", [False, True])
def test_step_fast_path_survives_a_rollout_in_between(ids):
    """A RandomRollout writes the env's MpeBuffers (act / ids of BOTH output sets) behind step()'s back: the short path's
    note of what each set's `act` holds must not survive that (round-3 ADVICE: step x3, enqueue, step, step stepped with the
    rollout's pool moves).  Sequence from the finding, against an env that never takes the short path."""
    from multiagent_particle_envs_amd.rollout import RandomRollout
    B = 640
    fast = mpe.make_env("simple_spread", batch_size=B, seed=5)
    slow = mpe.make_env("simple_spread", batch_size=B, seed=5)
    A = fast.n
    act = torch.zeros((A, B, 5), device="cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    rolls = [RandomRollout(e, episode_len=0, pool=4, action_ids=ids) for e in (fast, slow)]

    def both(tag):
        act.copy_(torch.rand((A, B, 5), device="cuda", generator=g))
        of, rf = fast.step(act)[:2]
        os_, rs = slow.step(act.clone())[:2]
        assert torch.equal(fast.world.pos, slow.world.pos) and torch.equal(fast.world.vel, slow.world.vel), tag
        for i in range(A):
            assert torch.equal(of[i], os_[i]) and torch.equal(rf[i], rs[i]), (tag, i)
    for t in range(3):
        both(("armed", t))
    assert id(act) in fast._fast_acts
    for n in (1, 2, 3):
        for r in rolls:
            r.enqueue(n)
        both(("after enqueue", n, 0))
        both(("after enqueue", n, 1))
        both(("after enqueue", n, 2))
    for r in rolls:
        r.fused(3)
    for t in range(3):
        both(("after fused", t))
    gr = [r.capture(4) for r in rolls]
    for k in range(2):
        for x in gr:
            x.replay()
        for t in range(3):
            both(("after replay", k, t))


@pytest.mark.parametrize("name,kw", [("simple_spread", {}), ("simple_tag", {}), ("simple_spread", {"num_agents": 20}),
                                     ("simple_reference", {}), ("simple_world_comm", {})])
def test_partial_fusion_python_observation_over_the_fused_step(name, kw):
    """A user who overrides only `observation` (extra features, another layout) keeps ONE launch for action decode,
    World.step and the built-in reward; the Python rows are evaluated on the post-step world (environment.py:92-97).
    Rewards / state bit-identical to the fully fused env, rows identical to what the generic path returns for the same
    subclass; reset() and the device-side auto-reset hand out the Python rows too."""
    B = 600
    Base = mpe.scenarios.load(name + ".py").Scenario

    class Mine(Base):
        def observation(self, agent, world):
            return torch.cat([Base.observation(self, agent, world) * 2.0, agent.state.p_pos.norm(dim=1, keepdim=True)], dim=1)
    sc = Mine()
    w = sc.make_world(batch_size=B, **kw)
    sc.reset_world(w)
    part = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation)
    part.scenario = sc
    assert part.fused and part._py_obs and not part._py_reward
    ref = mpe.make_env(name, batch_size=B, **kw)
    sc2 = Mine()
    w2 = sc2.make_world(batch_size=B, **kw)
    gen = mpe.MultiAgentEnv(w2, sc2.reset_world, sc2.reward, sc2.observation, fused=False)
    assert not gen.fused
    for i in range(part.n):
        assert part.observation_space[i].shape == (ref.observation_space[i].shape[0] + 1,) == gen.observation_space[i].shape
    for e, s_ in ((ref, ref.scenario), (gen, sc2)):
        e.world.pos.copy_(w.pos)
        e.world.vel.copy_(w.vel)
        if w.choice_i32 is not None:
            e.world.choice_i32.copy_(w.choice_i32)
            if hasattr(s_, "_apply"):
                s_._apply(e.world)
    if w.choice_i32 is not None and hasattr(sc, "_apply"):
        sc._apply(w)
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(3):
        acts = []
        for agent in part.agents:
            parts = []
            if agent.movable:
                parts.append(torch.nn.functional.one_hot(torch.randint(0, 5, (B,), device="cuda", generator=g), 5).float())
            if not agent.silent:
                parts.append(torch.nn.functional.one_hot(torch.randint(0, w.dim_c, (B,), device="cuda", generator=g), w.dim_c).float())
            acts.append(torch.cat(parts, dim=1))
        o_p, r_p, d_p, _ = part.step(acts)
        o_r, r_r, _, _ = ref.step(acts)
        o_g, r_g, _, _ = gen.step(acts)
        assert torch.equal(part.world.pos, ref.world.pos) and torch.equal(part.world.vel, ref.world.vel)
        for i in range(part.n):
            assert torch.equal(r_p[i], r_r[i]) and not d_p[i].any()
            close(np_(r_p[i]), np_(r_g[i]), what="rew%d vs generic" % i)
            assert o_p[i].shape == (B, o_r[i].shape[1] + 1)
            close(np_(o_p[i][:, :-1]), np_(o_r[i]) * 2.0, what="obs%d vs fused rows" % i)
            close(np_(o_p[i]), np_(o_g[i]), what="obs%d vs generic" % i)
    # reset(): Python rows of the fresh state
    seeds = list(range(40, 40 + B))
    o_p, o_r = part.reset(seeds=seeds), ref.reset(seeds=seeds)
    for i in range(part.n):
        close(np_(o_p[i][:, :-1]), np_(o_r[i]) * 2.0, what="reset obs%d" % i)
        close(np_(o_p[i][:, -1]), np.linalg.norm(np_(part.world.agents[i].state.p_pos), axis=1), what="reset extra column %d" % i)
    # the device-side auto-reset at the horizon: the rows returned at that step are the new episode's first
    sc3 = Mine()
    w3 = sc3.make_world(batch_size=B, **kw)
    sc3.reset_world(w3)
    auto = mpe.MultiAgentEnv(w3, sc3.reset_world, sc3.reward, sc3.observation, max_episode_steps=2, auto_reset=True)
    assert auto.fused and auto._py_obs
    auto.reset()
    zero = [torch.zeros_like(a) for a in acts]
    auto.step(zero)
    before = auto.world.pos.clone()
    o, _, d, _ = auto.step(zero)
    assert all(x.all() for x in d) and not torch.equal(auto.world.pos, before)
    for i in range(auto.n):
        close(np_(o[i][:, -1]), np.linalg.norm(np_(auto.world.agents[i].state.p_pos), axis=1), what="auto-reset rows %d" % i)
    from multiagent_particle_envs_amd import _abi
    from multiagent_particle_envs_amd.rollout import RandomRollout
    with pytest.raises(_abi.MpeError):
        RandomRollout(part)


@pytest.mark.parametrize("name,kw", [("simple_spread", {}), ("simple_tag", {}), ("simple_spread", {"num_agents": 20}),
                                     ("simple_reference", {})])
def test_partial_fusion_python_reward_done_info_over_the_fused_step(name, kw):
    """A user who overrides only reward (or adds done / benchmark callbacks) keeps ONE launch for action decode,
    World.step and the observation rows; the Python callbacks run on the post-step world.  Equivalent callbacks must
    give what the fully fused env and the generic path give (environment.py:92-102: obs, reward, done, info per agent,
    then the shared-reward sum)."""
    B = 700
    Base = mpe.scenarios.load(name + ".py").Scenario

    class Mine(Base):
        def reward(self, agent, world):
            return Base.reward(self, agent, world) * 2.0 + 1.0

        def my_done(self, agent, world):
            return agent.state.p_pos[:, 0] > 0.5

        def my_info(self, agent, world):
            return agent.state.p_vel[:, 1]
    sc = Mine()
    w = sc.make_world(batch_size=B, **kw)
    sc.reset_world(w)
    part = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, sc.my_info, sc.my_done)
    assert part.fused and part._py_reward and part._py_done and part._py_info
    ref = mpe.make_env(name, batch_size=B, **kw)                     # fully fused, built-in reward
    sc2 = Mine()
    w2 = sc2.make_world(batch_size=B, **kw)
    gen = mpe.MultiAgentEnv(w2, sc2.reset_world, sc2.reward, sc2.observation, sc2.my_info, sc2.my_done, fused=False)
    for e in (ref, gen):
        e.world.pos.copy_(w.pos)
        e.world.vel.copy_(w.vel)
        if w.choice_i32 is not None:
            e.world.choice_i32.copy_(w.choice_i32)
            if hasattr(e.scenario if hasattr(e, "scenario") else sc2, "_apply"):
                (e.scenario if hasattr(e, "scenario") else sc2)._apply(e.world)
    if w.choice_i32 is not None:
        sc._apply(w)
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(3):
        acts = []
        for agent in part.agents:
            parts = []
            if agent.movable:
                parts.append(torch.nn.functional.one_hot(torch.randint(0, 5, (B,), device="cuda", generator=g), 5).float())
            if not agent.silent:
                parts.append(torch.nn.functional.one_hot(torch.randint(0, w.dim_c, (B,), device="cuda", generator=g), w.dim_c).float())
            acts.append(torch.cat(parts, dim=1))
        o_p, r_p, d_p, i_p = part.step(acts)
        o_r, r_r, _, _ = ref.step(acts)
        o_g, r_g, d_g, i_g = gen.step(acts)
        assert torch.equal(part.world.pos, ref.world.pos) and torch.equal(part.world.vel, ref.world.vel)
        n = part.n
        scale = n if part.shared_reward else 1      # shared reward = sum over agents of (2 r_i + 1) = 2 * sum + n
        for i in range(n):
            assert torch.equal(o_p[i], o_r[i])
            close(np_(r_p[i]), np_(r_r[i]) * 2.0 + scale, what="rew%d vs fused" % i)
            close(np_(r_p[i]), np_(r_g[i]), what="rew%d vs generic" % i)
            assert torch.equal(d_p[i], d_g[i]) and torch.equal(d_p[i], part.world.agents[i].state.p_pos[:, 0] > 0.5)
            assert torch.equal(i_p["n"][i], i_g["n"][i])
