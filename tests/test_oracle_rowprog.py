"""The row-program oracle (oracle/rowprog.py) against the reference's goldens, and the peephole pass against it -- CPU only.

The nine shipped scenarios exist twice in this package: as fused kernels and as specs (rowspec.builtin_specs -> row programs).
The kernels are held to the reference by the GPU parity tests; these tests hold the SPECS to it without a GPU: each scenario's
program, evaluated by the NumPy restatement of the op table, must reproduce the observations and rewards the reference itself
recorded (tests/golden/*.npz) from the post-step states it recorded.  Tolerance 1e-7: an op carries its constants (colours, coefficients) as
float32 words -- 0.9f differs from 0.9 by 2.4e-8 --; everything else is fp64 against fp64 and agrees to 1e-12."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_rowspec as tr  # noqa: E402
from multiagent_particle_envs_amd import rowspec  # noqa: E402
from oracle import rowprog  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def oracle_of(env, dtype=np.float64):
    p = env._prog
    o = rowprog.from_program(p.struct, p.ops_host, p.n_ops, env._desc, dtype=dtype)
    o.size = np.asarray([e.size for e in env.world.entities], dtype)        # the Python floats (the descriptor holds float32)
    return o


def replay(name, g, scenario_kw=None, fuse=True):
    keep, rowspec.FUSE = rowspec.FUSE, fuse
    try:
        env = tr.make_spec_env(name, 4, device="cpu", scenario_kw=scenario_kw)
    finally:
        rowspec.FUSE = keep
    orc = oracle_of(env)
    A = orc.A
    T = g["pos"].shape[0]
    choice = g["choice"].T if "choice" in g.files and g["choice"].shape[1] else None
    worst = 0.0
    for t in range(T):
        comm = np.stack([g["c%d" % i][t] for i in range(A)]) if "c0" in g.files else None
        obs = orc.observe(g["pos"][t], g["vel"][t], comm, choice)
        rew = orc.rewards(g["pos"][t], g["vel"][t], comm, choice)
        for i in range(A):
            assert obs[i].shape == g["obs%d" % i][t].shape, (name, i, obs[i].shape, g["obs%d" % i][t].shape)
            worst = max(worst, float(np.abs(obs[i] - g["obs%d" % i][t]).max()))
            ref = g["rew"][t][:, i]
            worst = max(worst, float((np.abs(rew[i] - ref) / np.maximum(1.0, np.abs(ref))).max()))
    return worst, env._prog.n_ops


@pytest.mark.parametrize("name", tr.NINE)
@pytest.mark.parametrize("fuse", [True, False])
def test_the_nine_scenarios_as_specs_reproduce_the_reference_goldens(name, fuse):
    """Every op list -- one op per spec call, and after the peephole pass -- evaluates to the reference's own rows and rewards."""
    f = name if name in ("simple", "simple_spread", "simple_tag") else "f3_" + name
    g = np.load(os.path.join(GOLD, f + ".npz"))
    worst, n_ops = replay(name, g, fuse=fuse)
    assert worst <= 1e-7, (name, fuse, worst, n_ops)


SHAPES = sorted(f[len("shape_"):-4] for f in os.listdir(GOLD) if f.startswith("shape_"))


@pytest.mark.parametrize("shape", SHAPES)
def test_team_sizes_as_specs_reproduce_the_reference_goldens(shape):
    """simple_adversary / simple_world_comm at the 19 team sizes the reference was recorded at: the shapes that step through their
    row program (no kernel of their own) -- the program's value is the reference's."""
    g = np.load(os.path.join(GOLD, "shape_%s.npz" % shape))
    name, a, b = shape.rsplit("_", 2)
    kw = {"num_agents": int(a), "num_adversaries": int(b)} if name == "simple_adversary" else \
        {"num_good_agents": int(a) - int(b), "num_adversaries": int(b)}
    worst, _ = replay(name, g, scenario_kw=kw)
    assert worst <= 1e-7, (shape, worst)


@pytest.mark.parametrize("seed", range(1, 13))
def test_the_peephole_pass_does_not_change_what_a_random_program_computes(seed):
    """Random scenarios (tests/test_rowspec.py: _RandomScenario): the op list with range / grid forms and the one without evaluate to
    the same rows, rewards and dones on random states -- the fuser is value-preserving, decided on the CPU."""
    def build(fuse):
        keep, rowspec.FUSE = rowspec.FUSE, fuse
        try:
            sc = tr._RandomScenario(seed)
            w = sc.make_world(batch_size=4, device="cpu")
            import multiagent_particle_envs_amd as mpe
            env = mpe.MultiAgentEnv(w, sc.reset_world, None, None, compile_program=False)
        finally:
            rowspec.FUSE = keep
        return env, sc
    (e1, sc), (e2, _) = build(True), build(False)
    assert e1._prog.n_ops <= e2._prog.n_ops
    o1, o2 = oracle_of(e1), oracle_of(e2)
    rs = np.random.RandomState(100 + seed)
    B, A, E = 300, sc.A, sc.A + sc.Lm
    pos = rs.uniform(-1, 1, (B, E, 2))
    pos[::3] *= 0.3                                   # crowded worlds: contacts
    vel = rs.uniform(-1, 1, (B, A, 2))
    comm = np.eye(3)[rs.randint(0, 3, (A, B))] * np.array([0.0 if a.silent else 1.0 for a in e1.world.agents])[:, None, None]
    choice = np.stack([rs.randint(0, sc.Lm, B), rs.randint(0, sc.A, B)])
    for x, y in zip(o1.observe(pos, vel, comm, choice), o2.observe(pos, vel, comm, choice)):
        assert x.shape == y.shape and np.array_equal(x, y)
    for x, y in zip(o1.rewards(pos, vel, comm, choice), o2.rewards(pos, vel, comm, choice)):
        assert np.abs(x - y).max() <= 1e-12 * max(1.0, np.abs(y).max())
    for x, y in zip(o1.dones(pos, vel, comm, choice), o2.dones(pos, vel, comm, choice)):
        assert np.array_equal(x, y)
    assert any(d.any() for d in o1.dones(pos, vel, comm, choice)) or not e1._prog.has_done


def test_the_example_scenarios_two_descriptions_agree_on_the_cpu():
    """examples/corral.py describes its rows / rewards / done condition as specs AND computes them with torch callbacks: the program's
    value (this oracle, fp64) against the callbacks' (torch, fp32 tensors on the CPU) on random states -- without a GPU."""
    import torch
    B = 400
    env = tr.corral_env(B, device="cpu", arena=0.9)
    sc, w = env.scenario, env.world
    orc = oracle_of(env)
    rs = np.random.RandomState(8)
    pos = rs.uniform(-1.05, 1.05, (B, 6, 2))
    pos[::4] *= 0.25
    vel = rs.uniform(-1, 1, (B, 3, 2))
    choice = rs.randint(0, 3, (1, B))
    w.pos.copy_(torch.as_tensor(pos, dtype=torch.float32).permute(1, 2, 0))
    w.vel.copy_(torch.as_tensor(vel, dtype=torch.float32).permute(1, 2, 0))
    w.choice_i32.copy_(torch.as_tensor(choice, dtype=torch.int32))
    p32, v32 = w.pos.permute(2, 0, 1).double().numpy(), w.vel.permute(2, 0, 1).double().numpy()      # the fp32 state both sides see
    obs, rew, done = orc.observe(p32, v32, None, choice), orc.rewards(p32, v32, None, choice), orc.dones(p32, v32, None, choice)
    near = orc.reward_guard(p32, 1e-5, choice) | orc.done_guard(p32, 1e-5, v32, None, choice)
    assert near.mean() < 0.05
    for i, agent in enumerate(w.agents):
        o = sc.observation(agent, w).double().numpy()
        assert o.shape == obs[i].shape and np.abs(o - obs[i]).max() <= 1e-6, (i, np.abs(o - obs[i]).max())
        r = sc.reward(agent, w).double().numpy()
        e = np.abs(r - rew[i]) / np.maximum(1.0, np.abs(rew[i]))
        assert e[~near].max() <= 1e-5, (i, e[~near].max())
        d = sc.done(agent, w).numpy()
        assert np.array_equal(d[~near], done[i][~near]) and d.any()
