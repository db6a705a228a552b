"""ObsSpec / RewardSpec row programs (rowspec.py, csrc/mpe_rows.hip): a scenario DESCRIBES its observation rows and reward
terms, `mpe_rows` interprets them -- two launches per step for any scenario.

  not gpu   the nine built-ins compile to programs whose row widths equal the kernels' layout (mpe_fill_obs_layout) and pass
            mpe_rows_validate; malformed programs are refused with the op named
  -m gpu    the nine built-ins expressed as specs are BIT-IDENTICAL to their fused kernels (rows, rewards, state) over
            seeded episodes with resets; a custom scenario written only as specs matches its own torch callbacks (generic
            path) and steps through env.step, reset, auto-reset, GraphedStep
"""
import ctypes as C

import numpy as np
import pytest
import torch

import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _abi, rowspec
from multiagent_particle_envs_amd.core import World, Agent, Landmark
from multiagent_particle_envs_amd.scenario import BaseScenario

NINE = ["simple", "simple_spread", "simple_tag", "simple_adversary", "simple_push", "simple_speaker_listener", "simple_reference",
        "simple_crypto", "simple_world_comm"]


def spec_scenario(name):
    """The built-in scenario `name` with its callbacks REPLACED by specs (no kernel of its own: kind = None)."""
    Base = mpe.scenarios.load(name + ".py").Scenario

    class AsSpecs(Base):
        kind = None

        def _specs(self, world):
            if getattr(self, "_cache", None) is None or self._cache[0] is not world:
                self._cache = (world, rowspec.builtin_specs(name, world))
            return self._cache[1]

        def obs_spec(self, agent, world):
            return self._specs(world)[0][world.agents.index(agent)]

        def reward_spec(self, agent, world):
            return self._specs(world)[1][world.agents.index(agent)]

        def regions(self, world):
            return self._specs(world)[2]
    return AsSpecs()


def make_spec_env(name, B, device=None, seed=0, scenario_kw=None, **kw):
    sc = spec_scenario(name)
    w = sc.make_world(batch_size=B, device=device, **(scenario_kw or {}))
    w.seed = seed
    if w.pos.is_cuda:
        sc.reset_world(w)          # as make_env does (the reference's make_world ends with reset_world): same episode numbering
    kw.setdefault("compile_program", False)      # interpreted unless a test compiles it in (a cached image would attach itself)
    env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, **kw)
    env.scenario = sc
    return env


@pytest.mark.parametrize("name", NINE)
def test_builtin_specs_compile_to_the_kernels_row_layout(name):
    env = make_spec_env(name, 4, device="cpu")
    assert env.fused and env._prog is not None and env._kind == _abi.MPE_SCN_GENERIC
    ref = mpe.scenarios.load(name + ".py").Scenario()
    w = ref.make_world(batch_size=4, device="cpu")
    d = w.scenario_desc(ref.kind, getattr(ref, "num_adversaries", 0))
    A = len(w.agents)
    assert [int(env._desc.obs_off[i]) for i in range(A + 1)] == [int(d.obs_off[i]) for i in range(A + 1)]
    assert [s.shape[0] for s in env.observation_space] == [int(d.obs_off[i + 1] - d.obs_off[i]) for i in range(A)]
    assert env._prog.n_ops < 400


def test_malformed_programs_are_refused_with_the_op_named():
    sc = mpe.scenarios.load("simple_adversary.py").Scenario()
    w = sc.make_world(batch_size=2, device="cpu")
    obs, rew, _ = rowspec.builtin_specs("simple_adversary", w)
    good = rowspec.RowProgram(w, obs, rew)
    d = w.scenario_desc(_abi.MPE_SCN_GENERIC)
    dd = _abi.MpeScenarioDesc()
    C.memmove(C.byref(dd), C.byref(d), C.sizeof(d))
    off = 0
    for i, wd in enumerate(good.widths):
        dd.obs_off[i] = off
        off += wd
    dd.obs_off[len(good.widths)] = off
    good.validate(dd)
    dd.obs_off[1] += 1                                   # a row width that is not what the program emits
    with pytest.raises(_abi.MpeError, match="emits"):
        good.validate(dd)
    dd.obs_off[1] -= 1
    bad = rowspec.ObsSpec(w, w.agents[0])
    bad.ops.append(rowspec._op(_abi.MPE_ROW_OBS_REL, 77))          # entity 77 of 5
    bad.width = 2
    with pytest.raises(_abi.MpeError, match="entity 77"):
        p = rowspec.RowProgram(w, [bad] + obs[1:], rew)
        dd.obs_off[1], dd.obs_off[2], dd.obs_off[3] = 2, 2 + good.widths[1], 2 + good.widths[1] + good.widths[2]
        p.validate(dd)
    r = rowspec.RewardSpec(w, w.agents[0])
    r.ops.append(rowspec._op(_abi.MPE_ROW_R_STORE, 2))             # agent 0's program storing agent 2's reward
    with pytest.raises(_abi.MpeError, match="stores agent 2"):
        p = rowspec.RowProgram(w, obs, [r] + rew[1:])
        for i in range(4):
            dd.obs_off[i] = sum(good.widths[:i])
        p.validate(dd)
    r = rowspec.RewardSpec(w, w.agents[0])
    r.ops.append(rowspec._op(_abi.MPE_ROW_R_LOAD, 9))              # slot 9 of 8
    with pytest.raises(_abi.MpeError, match="slot 9"):
        p = rowspec.RowProgram(w, obs, [r] + rew[1:])
        for i in range(4):
            dd.obs_off[i] = sum(good.widths[:i])
        p.validate(dd)
    with pytest.raises(_abi.MpeError, match="not an entity"):
        rowspec.ObsSpec(w, w.agents[0]).rel(Landmark())


# ---- a scenario nobody wrote a kernel for, described by specs only: examples/corral.py ------------------------------------------
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from corral import Scenario as Corral  # noqa: E402


def corral_env(B, fused=None, device=None, arena=None, **kw):
    sc = Corral()
    sc.arena = arena
    w = sc.make_world(batch_size=B, device=device) if device else sc.make_world(batch_size=B)
    w.seed = 3
    kw.setdefault("compile_program", False)
    env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, fused=fused, **kw)
    env.scenario = sc
    return env


def rand_actions(env, rs, B):
    acts = []
    for agent in env.agents:
        parts = []
        if agent.movable:
            parts.append(np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=B)])
        if not agent.silent:
            parts.append(np.eye(env.world.dim_c, dtype=np.float32)[rs.randint(0, env.world.dim_c, size=B)])
        acts.append(torch.as_tensor(np.concatenate(parts, axis=1)).cuda())
    return acts


def as_tuple(env, acts, B):
    """The per-agent [move | utterance] rows as the batched pair (moves [A,B,5], utterances [A,B,dim_c]); rows an agent does not
    have (a speaker that cannot move, a silent listener) are junk on purpose: they must not matter."""
    dc = env.world.dim_c
    moves = torch.full((env.n, B, 5), 0.37, device="cuda")
    words = torch.full((env.n, B, dc), 0.91, device="cuda")
    for i, agent in enumerate(env.agents):
        k = 0
        if agent.movable:
            moves[i] = acts[i][:, :5]
            k = 5
        if not agent.silent:
            words[i] = acts[i][:, k:k + dc]
    return moves.contiguous(), words.contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["simple_speaker_listener", "simple_reference", "simple_crypto", "simple_world_comm"])
@pytest.mark.parametrize("program", [False, True])
def test_moves_and_utterances_as_one_pair_of_tensors_equal_the_per_agent_rows(name, program):
    """env.step((moves, utterances)) -- two device tensors, one staging launch -- against the reference's per-agent
    [move | utterance] rows (two small copies per agent): rows, rewards, state, comm state to the bit; fused kernels and programs."""
    B = 2048
    mk = (lambda: make_spec_env(name, B, seed=2)) if program else (lambda: mpe.make_env(name, batch_size=B, seed=2))
    a, b = mk(), mk()
    rs = np.random.RandomState(4)
    a.reset(), b.reset()
    for t in range(6):
        acts = rand_actions(a, rs, B)
        oa, ra, _, _ = a.step(acts)
        ob, rb, _, _ = b.step(as_tuple(b, acts, B))
        assert torch.equal(a.world.pos, b.world.pos) and torch.equal(a.world.vel, b.world.vel), t
        assert torch.equal(a._comm, b._comm), t
        assert all(torch.equal(x, y) for x, y in zip(oa + ra, ob + rb)), t
        for x, y in zip(a.world.agents, b.world.agents):
            if not x.silent:
                assert torch.equal(x.state.c, y.state.c)
    with pytest.raises(_abi.MpeError, match="two contiguous float32 device tensors"):
        b.step((torch.zeros(2, 2), torch.zeros(2, 2)))
    # the same pair of tensors again, rewritten in place: step()'s short path (one staging launch + the step), same results
    pair = as_tuple(b, rand_actions(a, rs, B), B)
    for t in range(5):
        acts = rand_actions(a, rs, B)
        fresh = as_tuple(b, acts, B)
        pair[0].copy_(fresh[0]), pair[1].copy_(fresh[1])
        oa, ra, _, _ = a.step(acts)
        ob, rb, _, _ = b.step(pair)
        assert all(torch.equal(x, y) for x, y in zip(oa + ra, ob + rb)) and torch.equal(a._comm, b._comm), ("short path", t)
        if t == 2:
            a.reset(), b.reset()
    assert id(pair[0]) in b._fast_acts


@pytest.mark.gpu
@pytest.mark.parametrize("name", NINE)
@pytest.mark.parametrize("B", [1000, 8192])
def test_builtins_as_specs_are_bit_identical_to_their_fused_kernels(name, B):
    """Same seed, same moves / words, resets in between: rows, rewards, dones and state of the two-launch program path ==
    the one-launch fused kernel's, to the bit (B = 1000: a ragged last workgroup and unaligned rows)."""
    fused = mpe.make_env(name, batch_size=B, seed=5)
    prog = make_spec_env(name, B, seed=5)                 # mpe_step_rows: World.step + the programs in ONE launch
    rowspec.FUSE = False
    try:
        prog2 = make_spec_env(name, B, seed=5)            # mpe_world_step + mpe_rows: two launches, and one op per spec call (no range forms)
    finally:
        rowspec.FUSE = True
    prog2.two_launch_program = True
    progc = make_spec_env(name, B, seed=5)                # the program COMPILED IN (mpe_rows_load_image): straight-line code
    assert progc.compile_program() and progc.program_compiled and not prog.program_compiled
    assert prog2._prog.n_ops >= prog._prog.n_ops
    assert fused.fused and fused._prog is None and prog.fused and prog._prog is not None
    rs = np.random.RandomState(B)
    of, op, op2, oc = fused.reset(), prog.reset(), prog2.reset(), progc.reset()
    for i in range(fused.n):
        assert torch.equal(of[i], op[i]) and torch.equal(of[i], op2[i]) and torch.equal(of[i], oc[i]), ("reset", i)
    for t in range(9):
        if t == 2:                                   # crowd the worlds: contacts, boundary penalties
            for e in (fused, prog, prog2, progc):
                e.world.pos.mul_(0.35)
        if t == 6:
            of, op, op2, oc = fused.reset(), prog.reset(), prog2.reset(), progc.reset()
            for i in range(fused.n):
                assert torch.equal(of[i], op[i]) and torch.equal(of[i], op2[i]) and torch.equal(of[i], oc[i]), ("second reset", i)
        act = rand_actions(fused, rs, B)
        of, rf, df, _ = fused.step(act)
        op, rp, dp, _ = prog.step(act)
        op2, rp2, _, _ = prog2.step(act)
        oc, rc, dc, _ = progc.step(act)
        for e in (prog, prog2, progc):
            assert torch.equal(fused.world.pos, e.world.pos) and torch.equal(fused.world.vel, e.world.vel), t
        for i in range(fused.n):
            assert torch.equal(of[i], op[i]), (name, "obs", t, i, float((of[i] - op[i]).abs().max()))
            assert torch.equal(rf[i], rp[i]), (name, "rew", t, i, float((rf[i] - rp[i]).abs().max()))
            assert torch.equal(df[i], dp[i])
            assert torch.equal(of[i], op2[i]) and torch.equal(rf[i], rp2[i]), (name, "two launches", t, i)
            assert torch.equal(of[i], oc[i]), (name, "compiled obs", t, i, float((of[i] - oc[i]).abs().max()))
            assert torch.equal(rf[i], rc[i]) and torch.equal(df[i], dc[i]), (name, "compiled rew", t, i, float((rf[i] - rc[i]).abs().max()))
    assert progc.program_compiled


@pytest.mark.gpu
def test_moves_as_one_tensor_and_integer_ids_on_the_program_path():
    B = 2048
    fused, prog = mpe.make_env("simple_spread", batch_size=B, seed=1), make_spec_env("simple_spread", B, seed=1)
    fused.reset(), prog.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    act = torch.rand((3, B, 5), device="cuda", generator=g)
    for t in range(4):                                # the same preallocated tensor again: the step's short path
        act.copy_(torch.rand((3, B, 5), device="cuda", generator=g))
        of, rf = fused.step(act)[:2]
        op, rp = prog.step(act)[:2]
        assert all(torch.equal(a, b) for a, b in zip(of + rf, op + rp)), t
    assert id(act) in prog._fast_acts
    for e in (fused, prog):
        e.discrete_action_input = True
    ids = torch.randint(0, 5, (3, B), device="cuda", dtype=torch.int32, generator=g)
    of, rf = fused.step(ids)[:2]
    op, rp = prog.step(ids)[:2]
    assert all(torch.equal(a, b) for a, b in zip(of + rf, op + rp))


@pytest.mark.gpu
def test_custom_scenario_specs_against_its_own_torch_callbacks():
    """Corral has no kernel: as specs it runs World.step + mpe_rows, as torch callbacks the generic path; same worlds, same
    moves -- rows to 1e-6 (differences and copies), rewards to 1e-5 (torch's sqrt / exp vs the kernels'), state bit-identical."""
    B = 4096
    ps, pt = corral_env(B), corral_env(B, fused=False)
    assert ps._prog is not None and ps.fused and not pt.fused
    rs = np.random.RandomState(0)
    o1, o2 = ps.reset(), pt.reset()
    assert torch.equal(ps.world.choice_i32, pt.world.choice_i32) and len(set(ps.world.choice_i32[0].tolist())) == 3
    for i in range(3):
        assert o1[i].shape == (B, 24) and torch.allclose(o1[i], o2[i], atol=1e-6, rtol=0)
    for t in range(12):
        if t == 3:
            for e in (ps, pt):
                e.world.pos.mul_(0.3)
        if t == 8:
            m = torch.arange(B, device="cuda") % 3 == 0
            ps.reset(mask=m), pt.reset(mask=m)
        act = rand_actions(ps, rs, B)
        o1, r1, d1, _ = ps.step(act)
        o2, r2, d2, _ = pt.step(act)
        assert torch.equal(ps.world.pos, pt.world.pos) and torch.equal(ps.world.vel, pt.world.vel)
        for i in range(3):
            assert torch.allclose(o1[i], o2[i], atol=1e-6, rtol=0), (t, i)
            err = ((r1[i] - r2[i]).abs() / r2[i].abs().clamp(min=1.0)).max()
            assert float(err) <= 1e-5, (t, i, float(err))
            assert not d1[i].any()
    assert float(torch.stack(r1).min()) < -3.0           # contacts happened


@pytest.mark.gpu
def test_program_env_auto_reset_and_graphed_step():
    B = 512
    env = corral_env(B, max_episode_steps=4, auto_reset=True)
    ref = corral_env(B, fused=False, max_episode_steps=4, auto_reset=True)
    rs = np.random.RandomState(2)
    env.reset(), ref.reset()
    for t in range(1, 10):
        act = rand_actions(env, rs, B)
        o1, r1, d1, _ = env.step(act)
        o2, r2, d2, _ = ref.step(act)
        assert bool(d1[0].all()) == (t % 4 == 0) and torch.equal(d1[0], d2[0])
        for i in range(3):
            assert torch.allclose(o1[i], o2[i], atol=1e-6, rtol=0), (t, i)      # the finished worlds' rows are the new episode's first
    e2, e3 = corral_env(B), corral_env(B)
    e2.reset(), e3.reset()
    gs = mpe.GraphedStep(e2, rand_actions(e2, rs, B))
    for t in range(5):
        act = rand_actions(e2, rs, B)
        og, rg, _, _ = gs.step(act)
        oe, re_, _, _ = e3.step(act)
        assert all(torch.equal(a, b) for a, b in zip(og + rg, oe + re_)), t


@pytest.mark.gpu
def test_compiled_custom_scenario_equals_the_interpreted_one_and_falls_back_when_constants_change():
    """examples/corral.py compiled in: rows / rewards / state == the interpreted program's to the bit, also through the
    done_callback + auto_reset launch (mpe_episode_finish runs the image too) and a GraphedStep.  An edited constant (an
    agent resized) no longer equals what the image was compiled for: steps fall back to the interpreter -- still equal to a
    fresh interpreted env with the same edit -- until compile_program() is called again."""
    B = 3000
    kw = dict(max_episode_steps=6, auto_reset=True, done_callback=_strayed)
    a, b = corral_env(B, **kw), corral_env(B, **kw)
    assert a.compile_program() and a.program_compiled and not b.program_compiled
    rs = np.random.RandomState(3)
    a.reset(), b.reset()

    def same(t):
        act = rand_actions(a, rs, B)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(a.world.pos, b.world.pos) and torch.equal(a.world.vel, b.world.vel), t
        assert torch.equal(a.episode_step, b.episode_step) and torch.equal(a.world.choice_i32, b.world.choice_i32), t
        assert all(torch.equal(x, y) for x, y in zip(oa + ra + da, ob + rb + db)), t
    for t in range(14):
        if t == 4:
            for e in (a, b):
                e.world.pos.mul_(0.3)
        same(t)
    for e in (a, b):
        e.world.agents[1].size = 0.1234                # (a value tools/precompile_rows.py --tests does not put in the cache)
    same("edited")
    assert not a.program_compiled                      # the image is for the old size: interpreted now
    assert a.compile_program() and a.program_compiled  # a new image for the new constants
    for t in range(4):
        same(("recompiled", t))
    c, d = corral_env(B), corral_env(B)
    assert c.compile_program()
    c.reset(), d.reset()
    gs = mpe.GraphedStep(c, rand_actions(c, rs, B))
    for t in range(4):
        act = rand_actions(c, rs, B)
        og, rg, _, _ = gs.step(act)
        oe, re_, _, _ = d.step(act)
        assert all(torch.equal(x, y) for x, y in zip(og + rg, oe + re_)), t


@pytest.mark.gpu
@pytest.mark.parametrize("compiled", [False, True])
@pytest.mark.parametrize("horizon", [7, 1000])
def test_episodes_that_end_inside_the_step_launch_equal_step_plus_callback_plus_finish(compiled, horizon):
    """A done_spec (`arena`: an agent outside the square is done) + auto_reset: mpe_step_rows_episode -- step, done tests,
    restart of the finished worlds, their new rows in ONE launch -- against the same scenario with the condition as a torch
    done_callback (mpe_step_rows, the callback, mpe_episode_finish): rows, rewards, dones, counters, state, picks to the bit;
    with a short horizon both kinds of episode end occur, with a long one only the done tests."""
    B = 3000
    a = corral_env(B, arena=0.95, max_episode_steps=horizon, auto_reset=True)
    b = corral_env(B, max_episode_steps=horizon, auto_reset=True, done_callback=_strayed)
    assert a._prog.has_done and a._episode_in_launch and not b._prog.has_done and not b._episode_in_launch
    if compiled:
        assert a.compile_program() and b.compile_program()
    rs = np.random.RandomState(5)
    oa, ob = a.reset(), b.reset()
    assert all(torch.equal(x, y) for x, y in zip(oa, ob))
    ended = 0
    for t in range(1, 25):
        if t == 9:
            for e in (a, b):
                e.world.pos.mul_(0.3)
        act = rand_actions(a, rs, B)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(a.world.pos, b.world.pos) and torch.equal(a.world.vel, b.world.vel), t
        assert torch.equal(a.episode_step, b.episode_step) and torch.equal(a.world.choice_i32, b.world.choice_i32), t
        for i in range(3):
            assert torch.equal(da[i], db[i]), (t, i, int(da[i].sum()), int(db[i].sum()))
            assert torch.equal(oa[i], ob[i]), (t, i)
            assert torch.equal(ra[i], rb[i]), (t, i)
        ended += int(torch.stack(da).any(dim=0).sum())
    assert ended > B // 20
    assert a.program_compiled == compiled


@pytest.mark.gpu
def test_horizon_only_auto_reset_of_a_program_env_runs_in_the_step_launch():
    """No done condition at all: max_episode_steps + auto_reset of a row-program env is still one launch per step, equal to
    the separate launches (`finish_launch = False`: tick, masked reset, mpe_rows at the horizon)."""
    B = 2000
    a = corral_env(B, max_episode_steps=5, auto_reset=True)
    b = corral_env(B, max_episode_steps=5, auto_reset=True)
    b.finish_launch = False
    assert a._episode_in_launch and not b._episode_in_launch
    rs = np.random.RandomState(6)
    a.reset(), b.reset()
    for t in range(1, 14):
        act = rand_actions(a, rs, B)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert bool(da[0].all()) == (t % 5 == 0)
        assert torch.equal(a.world.pos, b.world.pos) and torch.equal(a.episode_step, b.episode_step), t
        assert all(torch.equal(x, y) for x, y in zip(oa + ra + da, ob + rb + db)), t


def test_done_programs_are_validated():
    sc = Corral()
    w = sc.make_world(batch_size=2, device="cpu")
    ag = w.agents
    obs = [sc.obs_spec(x, w) for x in ag]
    rew = [sc.reward_spec(x, w) for x in ag]
    ok = rowspec.RowProgram(w, obs, rew, done_specs=[rowspec.DoneSpec(w, ag[0]).outside(ag[0], 0.9), None,
                                                     rowspec.DoneSpec(w, ag[2]).dist(ag[2], w.landmarks[0]).done_if_lt(0.1).done_if_touching(ag[2], ag[0])])
    env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, compile_program=False)
    ok.validate(env._desc)
    assert ok.has_done and [ok.struct.done_begin[i] for i in range(4)] == [ok.n_ops - 8, ok.n_ops - 4, ok.n_ops - 4, ok.n_ops]
    assert not env._prog.has_done and all(env._prog.struct.done_begin[i] == 0 for i in range(65))
    bad = rowspec.DoneSpec(w, ag[0])
    bad.add_if_touching(ag[0], ag[1], -1.0)           # a reward term in a done program
    with pytest.raises(_abi.MpeError, match="belongs to reward programs"):
        rowspec.RowProgram(w, obs, rew, done_specs=[bad, None, None]).validate(env._desc)
    worse = sc.reward_spec(ag[0], w)
    worse.ops.append(rowspec._op(_abi.MPE_ROW_R_DONE_IF_GT, f0=1.0))
    with pytest.raises(_abi.MpeError, match="belongs to the done program"):
        rowspec.RowProgram(w, obs, [worse] + rew[1:]).validate(env._desc)


@pytest.mark.gpu
def test_a_cached_image_attaches_itself_and_a_missing_one_does_not_start_hipcc(tmp_path, monkeypatch):
    from multiagent_particle_envs_amd import _build
    warm = corral_env(64)
    assert warm.compile_program()                               # the cache holds Corral's image now
    assert corral_env(64, compile_program=None).program_compiled            # default policy: found, attached
    assert not corral_env(64, compile_program=False).program_compiled
    monkeypatch.setattr(_build, "ROWS_CACHE", str(tmp_path))    # an empty cache
    monkeypatch.setattr(_build, "_hipcc", lambda: (_ for _ in ()).throw(AssertionError("hipcc must not run")))
    assert not corral_env(64, compile_program=None).program_compiled        # nothing cached: interpreted, no compiler
    monkeypatch.undo()
    e = corral_env(64, compile_program=True)
    assert e.program_compiled
    e.world.agents[0].size = 0.07                               # policy True: a new image with the new constants at the next step
    e.reset()
    e.step(rand_actions(e, np.random.RandomState(0), 64))
    assert e.program_compiled


@pytest.mark.gpu
def test_an_image_of_another_program_is_refused():
    a, b = corral_env(64), make_spec_env("simple_spread", 64)
    from multiagent_particle_envs_amd import _build
    image = _build.compile_rows_image(a._prog.static_source(a._desc))
    buf = C.create_string_buffer(image, len(image))
    rc = _abi.lib().mpe_rows_load_image(C.byref(b._desc), b._prog.ref, b._prog.ops_host, buf, len(image))
    assert rc == -1 and b"compiled for another program" in _abi.lib().mpe_last_error()      # MPE_EINVAL
    assert not b.program_compiled
    with pytest.raises(_abi.MpeError, match="not through a row program"):
        mpe.make_env("simple_spread", batch_size=64).compile_program()


def test_static_source_and_image_of_a_program_on_the_cpu():
    """The generator and the compile need no GPU: the header names the kernels by a hash of (dims, tables, ops, waves), the
    code object holds the four entry points under that name; an edited constant gives another name."""
    env = make_spec_env("simple_tag", 4, device="cpu")
    src = env._prog.static_source(env._desc)
    name = [l.split()[2] for l in src.splitlines() if l.startswith("#define MPE_ROWS_STATIC_NAME")][0]
    assert name.startswith("mpe_rows_") and len(name) == 9 + 16
    for key in ("MPE_ROWS_STATIC_DIMS", "MPE_ROWS_STATIC_TABLES", "MPE_ROWS_STATIC_OPS", "MPE_ROWS_STATIC_WAVES_STEP", "MPE_ROWS_STATIC_WAVES_ROWS",
                "MPE_ROWS_STATIC_LDS_STEP", "MPE_ROWS_STATIC_LDS_ROWS", "MPE_ROWS_STATIC_OCC_STEP", "MPE_ROWS_STATIC_OCC_ROWS"):
        assert "#define %s " % key in src
    assert src.count("{") - 3 == env._prog.n_ops + 1          # one {a, b, c, d} per op (+ the dims' region pair)
    from multiagent_particle_envs_amd import _build
    image = _build.compile_rows_image(src)
    for suffix in ("_s", "_r", "_e", "_l", "_m"):
        assert (name + suffix).encode() in image
    assert _build.compile_rows_image(src) == image            # cached by content
    # the five entry points of the image keep their registers: no scratch memory (a spilling image would be slower than the interpreter)
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin/"
    if os.path.exists(llvm + "clang-offload-bundler"):
        with tempfile.TemporaryDirectory() as tmp:
            open(os.path.join(tmp, "i.hsaco"), "wb").write(image)
            subprocess.check_call([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + os.path.join(tmp, "i.hsaco"),
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + os.path.join(tmp, "i.elf")])
            notes = subprocess.check_output([llvm + "llvm-readelf", "--notes", os.path.join(tmp, "i.elf")]).decode()
        import re
        kernels = re.findall(r"\.name:\s+(%s_\w+)\n.*?\.private_segment_fixed_size:\s+(\d+)" % name, notes, flags=re.S)
        assert len(kernels) == 5 and all(int(sz) == 0 for _, sz in kernels), kernels
    env.world.agents[0].size = 0.2
    env.refresh_constants()
    other = env._prog.static_source(env._desc)
    assert [l for l in other.splitlines() if "STATIC_NAME" in l] != [l for l in src.splitlines() if "STATIC_NAME" in l]
    assert not env._prog.image_active(env._desc)


def test_a_program_of_too_many_ops_stays_interpreted():
    env = make_spec_env("simple_adversary", 4, device="cpu", scenario_kw={"num_agents": 30, "num_adversaries": 9})
    assert env._prog.n_ops > 512
    with pytest.raises(_abi.MpeError, match="stay interpreted"):
        env._prog.static_source(env._desc)


class _RandomScenario(BaseScenario):
    """A scenario drawn from a seed: 2-6 agents, 1-5 landmarks, two per-world picks, utterances; every agent's row, reward and
    done condition a random composition of the spec vocabulary (the peephole fuser and the compiled form must not care)."""

    landmark_range = 0.9
    device_reset = True        # (reset_world below is reset_uniform: restarts may be drawn on the device)

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.A, self.Lm = int(self.rs.randint(2, 7)), int(self.rs.randint(1, 6))
        self.plan = None

    def make_world(self, batch_size=1, device=None):
        rs = self.rs
        world = World(batch_size, device)
        world.dim_c = 3
        world.choice_pops = [self.Lm, self.A]
        world.agents = [Agent() for _ in range(self.A)]
        for i, a in enumerate(world.agents):
            a.name, a.collide, a.silent = "agent %d" % i, bool(rs.rand() < 0.8), bool(i % 2)
            a.size, a.accel, a.max_speed = float(rs.uniform(0.04, 0.15)), float(rs.uniform(2.0, 5.0)), float(rs.uniform(0.8, 1.5))
        world.landmarks = [Landmark() for _ in range(self.Lm)]
        for l in world.landmarks:
            l.collide, l.movable, l.size = bool(rs.rand() < 0.5), False, float(rs.uniform(0.03, 0.2))
        world.allocate()
        return world

    def reset_world(self, world, mask=None, seeds=None):
        idx = world.reset_uniform(self.landmark_range, mask, choices=[self.Lm, self.A], seeds=seeds)
        if world.choice_i32 is not None:
            for k in range(2):
                world.choice_i32[k].copy_(World.merge_choice(world.choice_i32[k].long(), idx[:, k].to(world.device), mask).to(torch.int32))

    def _plan(self, world):
        if self.plan is None:
            rs, ag, lm = self.rs, world.agents, world.landmarks
            ents = world.entities
            self.plan = []
            for a in ag:
                o = rowspec.ObsSpec(world, a)
                for _ in range(int(rs.randint(2, 7))):
                    k = int(rs.randint(0, 8))
                    if k == 0:
                        o.vel().pos()
                    elif k == 1:
                        for e in (lm if rs.rand() < 0.5 else lm[::-1]):      # a run the fuser can take, or a reversed one it cannot
                            o.rel(e)
                    elif k == 2:
                        for b in ag:
                            if b is not a:
                                o.rel(b)
                        for b in ag:
                            if b is not a:
                                o.vel(b)
                    elif k == 3:
                        o.rel_pick(0, lm).onehot(0, self.Lm, 0.25, 0.75)
                    elif k == 4:
                        o.comm(ag[int(rs.randint(0, self.A))])
                    elif k == 5:
                        o.const(float(rs.uniform(-1, 1)), float(rs.uniform(-1, 1)), float(rs.uniform(-1, 1)))
                    elif k == 6:
                        o.rel(ents[int(rs.randint(0, len(ents)))])
                    else:
                        o.onehot(1, self.A, 0.0, 1.0)
                r = rowspec.RewardSpec(world, a)
                for _ in range(int(rs.randint(1, 6))):
                    k = int(rs.randint(0, 7))
                    if k == 0:
                        for l in lm:
                            r.min_dist(ag, l).add(float(rs.choice([-1.0, -0.5])))
                    elif k == 1:
                        for b in ag:
                            if b is not a:
                                r.add_if_touching(b, a, float(rs.choice([-1.0, -3.0])))
                    elif k == 2:
                        r.min_dist2_from(a, lm).add(-0.25, acc=1)
                    elif k == 3:
                        r.dist2_pick(a, 0, lm).sqrt().save(3).load(3).add(-1.0)
                    elif k == 4:
                        r.bound(a, 0).add(-1.0).bound(a, 1).add(-1.0)
                    elif k == 5:
                        for b in ag:
                            for l in lm:
                                r.add_if_touching(b, l, 0.5, acc=1)
                    else:
                        r.comm_sum(ag[0]).add(0.1).add_acc1()
                d = None
                if rs.rand() < 0.6:
                    d = rowspec.DoneSpec(world, a).outside(a, 0.97)
                    if rs.rand() < 0.5:
                        d.min_dist(ag, lm[0]).done_if_lt(0.02).done_if_touching(a, lm[-1])
                self.plan.append((o, r, d))
        return self.plan

    def obs_spec(self, agent, world):
        return self._plan(world)[world.agents.index(agent)][0]

    def reward_spec(self, agent, world):
        return self._plan(world)[world.agents.index(agent)][1]

    def done_spec(self, agent, world):
        return self._plan(world)[world.agents.index(agent)][2]


def _random_env(seed, B, fuse, **kw):
    sc = _RandomScenario(seed)
    w = sc.make_world(batch_size=B)
    w.seed = seed
    sc.reset_world(w)
    keep, rowspec.FUSE = rowspec.FUSE, fuse
    try:
        env = mpe.MultiAgentEnv(w, sc.reset_world, None, None, compile_program=False, **kw)
    finally:
        rowspec.FUSE = keep
    env.scenario = sc
    return env


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_programs_one_op_per_call_vs_range_forms_vs_compiled_in(seed):
    """Programs nobody wrote by hand: the same random scenario with one op per spec call, with the peephole pass (range / grid
    forms), and compiled in -- rows, rewards, dones, state, the episode ends inside the launch: all three to the bit."""
    B = 1500
    kw = dict(max_episode_steps=6, auto_reset=True)
    plain, fused, comp = _random_env(seed, B, False, **kw), _random_env(seed, B, True, **kw), _random_env(seed, B, True, **kw)
    assert plain._prog.n_ops >= fused._prog.n_ops and comp.compile_program()
    envs = (plain, fused, comp)
    rs = np.random.RandomState(seed)
    obs = [e.reset() for e in envs]
    for o in obs[1:]:
        assert all(torch.equal(x, y) for x, y in zip(obs[0], o))
    for t in range(10):
        if t == 3:
            for e in envs:
                e.world.pos.mul_(0.4)
        act = rand_actions(plain, rs, B)
        outs = [e.step(act) for e in envs]
        for e in envs[1:]:
            assert torch.equal(plain.world.pos, e.world.pos) and torch.equal(plain.episode_step, e.episode_step), (seed, t)
            assert torch.equal(plain.world.choice_i32, e.world.choice_i32), (seed, t)
        for k, o in enumerate(outs[1:]):
            for i in range(plain.n):
                assert torch.equal(outs[0][0][i], o[0][i]), (seed, t, "obs", k, i)
                assert torch.equal(outs[0][1][i], o[1][i]), (seed, t, "reward", k, i)
                assert torch.equal(outs[0][2][i], o[2][i]), (seed, t, "done", k, i)
    assert comp.program_compiled and not fused.program_compiled


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_programs_against_the_numpy_oracle_of_the_op_table(seed):
    """What the kernel computes for a program nobody wrote by hand, against oracle/rowprog.py (the op table restated in NumPy
    fp64) on the kernel's own post-step state: rows and rewards to 1e-5, dones exactly outside a 2e-6 band around their
    thresholds, contact-counting rewards outside the same band around contact."""
    from oracle import rowprog
    B = 2048
    env = _random_env(seed, B, True)
    p = env._prog
    orc = rowprog.from_program(p.struct, p.ops_host, p.n_ops, env._desc)
    rs = np.random.RandomState(50 + seed)
    env.reset()
    worst = 0.0
    for t in range(6):
        if t == 2:
            env.world.pos.mul_(0.35)
        obs, rew, done, _ = env.step(rand_actions(env, rs, B))
        w = env.world
        pos = w.pos.permute(2, 0, 1).double().cpu().numpy()                    # [B, E, 2]
        vel = w.vel.permute(2, 0, 1).double().cpu().numpy()
        comm = env._comm.double().cpu().numpy()                                # [A, B, dim_c]
        choice = w.choice_i32.cpu().numpy()                                    # [K, B]
        o64 = orc.observe(pos, vel, comm, choice)
        r64 = orc.rewards(pos, vel, comm, choice)
        d64 = orc.dones(pos, vel, comm, choice)
        ok_r = ~orc.reward_guard(pos, 2e-6, choice)          # (soft contacts leave crowded worlds hovering at contact distance)
        ok_d = ~orc.done_guard(pos, 2e-6, vel, comm, choice)
        assert ok_r.mean() > 0.9 and ok_d.mean() > 0.9
        for i in range(env.n):
            worst = max(worst, float(np.abs(obs[i].double().cpu().numpy() - o64[i]).max()))
            e = np.abs(rew[i].double().cpu().numpy() - r64[i]) / np.maximum(1.0, np.abs(r64[i]))
            worst = max(worst, float(e[ok_r].max()))
            assert np.array_equal(done[i].cpu().numpy()[ok_d], d64[i][ok_d]), (seed, t, i)
    assert worst <= 1e-5, (seed, worst)


def _strayed(agent, world):
    """A done callback: the agent left the arena (any world, any step)."""
    return (agent.state.p_pos.abs() > 0.95).any(dim=1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["simple_spread", "simple_tag", "simple_adversary", "simple_reference", "simple_world_comm", "corral"])
def test_done_callback_auto_reset_in_one_launch_equals_the_separate_launches(name):
    """done_callback + auto_reset: mpe_episode_finish (tick, finished worlds, masked reset_world, comm state, their rows -- one
    launch with a per-workgroup early-out) against round 3's sequence of separate launches (tick, mask reduction, masked
    mpe_reset, comm fill, full mpe_observe): same worlds, same moves -- rows, rewards, dones, counters, state, picks to the bit."""
    B = 3000

    def build(finish):
        if name == "corral":
            env = corral_env(B, max_episode_steps=7, auto_reset=True, done_callback=_strayed)
        else:
            env = mpe.make_env(name, batch_size=B, seed=4, max_episode_steps=7, auto_reset=True)
            env.done_callback = _strayed
            env._py_done = True
        env.finish_launch = finish
        return env
    new, old = build(True), build(False)
    assert new.fused and new._finish_program() is not None and old._finish_program() is None
    rs = np.random.RandomState(1)
    o1, o2 = new.reset(), old.reset()
    restarted = 0
    for t in range(1, 17):
        act = rand_actions(new, rs, B)
        o1, r1, d1, _ = new.step(act)
        o2, r2, d2, _ = old.step(act)
        assert torch.equal(new.world.pos, old.world.pos) and torch.equal(new.world._vel_all, old.world._vel_all), t
        assert torch.equal(new.episode_step, old.episode_step), t
        if new.world.choice_i32 is not None:
            assert torch.equal(new.world.choice_i32, old.world.choice_i32), t
        if new._comm is not None:
            assert torch.equal(new._comm, old._comm), t
        for i in range(new.n):
            assert torch.equal(o1[i], o2[i]), (name, t, i)
            assert torch.equal(r1[i], r2[i]) and torch.equal(d1[i], d2[i]), (name, t, i)
        fin = torch.stack(d1).any(dim=0)
        restarted += int(fin.sum())
        assert torch.equal(new.episode_step == 0, fin), t          # exactly the finished worlds restarted their count
    assert 0 < restarted < 16 * B          # some worlds strayed early, most steps most worlds did not finish


# ---- whose rows, whose restarts (round 4's review) ------------------------------------------------------------------------------
def test_callbacks_that_are_not_the_scenarios_own_are_not_replaced_by_its_specs():
    """The specs stand in for the scenario's OWN observation / reward.  `MultiAgentEnv(world, sc.reset_world, my_reward, my_obs)`
    with a spec-bearing scenario keeps the caller's callbacks (generic path); asked for fused=True it is refused."""
    sc = Corral()
    w = sc.make_world(batch_size=4, device="cpu")

    def my_obs(agent, world):
        return torch.zeros((world.batch_size, 3))

    def my_reward(agent, world):
        return torch.ones(world.batch_size)
    env = mpe.MultiAgentEnv(w, sc.reset_world, my_reward, my_obs)
    assert env._prog is None and not env.fused and [s.shape[0] for s in env.observation_space] == [3, 3, 3]
    env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, my_obs)
    assert env._prog is None and not env.fused
    with pytest.raises(_abi.MpeError, match="fused=True"):
        mpe.MultiAgentEnv(w, sc.reset_world, my_reward, my_obs, fused=True)
    # no callbacks at all although the scenario has some: the reference's env then yields empty rows and zero rewards -- so does this
    env = mpe.MultiAgentEnv(w, sc.reset_world, None, None)
    assert env._prog is None and not env.fused and [s.shape[0] for s in env.observation_space] == [0, 0, 0]
    # the scenario's own methods (or none at all, for a spec-only scenario): the program
    assert mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation)._prog is not None
    r = _RandomScenario(3)
    assert mpe.MultiAgentEnv(r.make_world(batch_size=4, device="cpu"), r.reset_world, None, None)._prog is not None


def test_device_side_restarts_only_where_reset_world_is_the_uniform_placement():
    """In-launch / finish-launch restarts draw `reset_uniform` and never call reset_callback: taken for the built-in scenarios'
    own reset_world and for scenarios that declare `device_reset = True`, in rng_mode 'device' -- a scenario with a reset_world
    of its own keeps it (masked reset_callback path)."""
    class FixedPosts(Corral):
        device_reset = False

        def reset_world(self, world, mask=None, seeds=None):
            Corral.reset_world(self, world, mask, seeds)       # ... and then something of its own

    kw = dict(max_episode_steps=5, auto_reset=True, compile_program=False)
    for cls, ok in ((Corral, True), (FixedPosts, False)):
        sc = cls()
        w = sc.make_world(batch_size=4, device="cpu")
        env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, **kw)
        assert env._prog is not None and env._device_restart_ok == ok and env._episode_in_launch == ok
        assert (env._finish_program() is not None) == ok
        w.rng_mode = "numpy"                                    # compatibility mode: the global np.random stream, never the device
        assert not env._device_restart_ok and not env._episode_in_launch and env._finish_program() is None
    # a subclass that overrides a built-in scenario's reset_world is on its own too
    Base = mpe.scenarios.load("simple_tag.py").Scenario

    class Mine(Base):
        def reset_world(self, world, mask=None, seeds=None):
            Base.reset_world(self, world, mask, seeds)
    for cls, ok in ((Base, True), (Mine, False)):
        sc = cls()
        w = sc.make_world(batch_size=4, device="cpu")
        env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, max_episode_steps=5, auto_reset=True)
        assert env._device_restart_ok == ok


class _Wooded(Corral):
    """corral with post 0 as a region: every entity kind asked whether it is inside (observer, another agent, landmarks)."""

    def regions(self, world):
        return rowspec.Regions([world.landmarks[0]], [])

    def obs_spec(self, agent, world):
        o = Corral.obs_spec(self, agent, world)
        o.in_region(0).in_region(0, world.agents[2]).in_region(0, world.landmarks[1]).in_region(0, world.landmarks[0])
        return o.in_region(0, world.landmarks[2])


@pytest.mark.gpu
@pytest.mark.parametrize("compiled", [False, True])
def test_in_region_of_any_entity_against_the_numpy_oracle(compiled):
    """`ObsSpec.in_region(region, ent)` accepts any entity; the kernel kept the agents' bits only (a landmark always read -1).
    Rows against oracle/rowprog.py on the kernel's own state, posts pulled together so that landmarks DO overlap the region."""
    from oracle import rowprog
    B = 2048
    sc = _Wooded()
    w = sc.make_world(batch_size=B)
    w.seed = 5
    sc.reset_world(w)
    env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, compile_program=False)
    if compiled:
        assert env.compile_program()
    p = env._prog
    orc = rowprog.from_program(p.struct, p.ops_host, p.n_ops, env._desc)
    rs = np.random.RandomState(9)
    env.reset()
    env.world.pos.mul_(0.12)
    inside = np.zeros(5)
    for t in range(3):
        obs, _, _, _ = env.step(rand_actions(env, rs, B))
        pos = w.pos.permute(2, 0, 1).double().cpu().numpy()
        vel = w.vel.permute(2, 0, 1).double().cpu().numpy()
        choice = w.choice_i32.cpu().numpy()
        o64 = orc.observe(pos, vel, np.zeros((3, B, 0)), choice)
        d = np.sqrt(((pos - pos[:, 3:4, :]) ** 2).sum(-1))                                              # [B, E]: distance to post 0 (entity 3)
        size = np.array([e.size for e in w.entities])
        near = np.abs(d - (size[None, :] + size[3])) < 2e-6                                             # strict-< band (fp32 vs fp64)
        for i in range(env.n):
            g, r = obs[i].double().cpu().numpy(), o64[i]
            assert np.abs(g[:, :-5] - r[:, :-5]).max() <= 1e-5
            ents = [i, 2, 4, 3, 5]
            for k, e in enumerate(ents):
                ok = ~near[:, e]
                assert np.array_equal(g[ok, -5 + k], r[ok, -5 + k]), (t, i, k)
                inside[k] += (r[:, -5 + k] > 0).mean()
    assert inside[3] == 3 * env.n and 0 < inside[2] / (3 * env.n) < 1 and 0 < inside[4] / (3 * env.n) < 1     # post 0 is inside itself; others: both answers occur


@pytest.mark.gpu
def test_done_programs_restart_through_reset_world_where_it_is_not_the_device_draw():
    """A scenario with a done_spec whose reset_world is its OWN (no `device_reset`): with auto_reset the worlds its done tests flag
    are restarted at that step through the masked reset_callback -- the scenario's placement, not the device's uniform one."""
    class Pens(Corral):
        device_reset = False
        arena = 0.95

        def reset_world(self, world, mask=None, seeds=None):
            Corral.reset_world(self, world, mask, seeds)
            keep = None if mask is None else ~torch.as_tensor(mask, device=world.device).bool()
            new = world.pos[:3] * 0.25                           # agents start near the middle: [-0.25, 0.25)^2
            world.pos[:3] = new if keep is None else torch.where(keep[None, None, :], world.pos[:3], new)
    B = 2048
    sc = Pens()
    w = sc.make_world(batch_size=B)
    w.seed = 11
    sc.reset_world(w)
    env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, max_episode_steps=1000, auto_reset=True, compile_program=False)
    assert env._prog.has_done and not env._device_restart_ok and not env._episode_in_launch
    env.reset()
    assert float(env.world.pos[:3].abs().max()) < 0.25
    act = torch.zeros((3, B, 5), device="cuda")
    act[:, :, 1] = 1.0                                            # everybody accelerates in +x: they all leave the arena
    restarted = 0
    for t in range(40):
        obs, rew, done, _ = env.step(act)
        fin = done[0] | done[1] | done[2]
        if bool(fin.any()):
            restarted += int(fin.sum())
            assert bool((env.episode_step[fin] == 0).all()) and bool((env.episode_step[~fin] > 0).all())
            assert float(env.world.pos[:3][:, :, fin].abs().max()) < 0.25          # ITS placement, at the step the test fired
            assert float(env.world.vel[:3][:, :, fin].abs().max()) == 0.0
    assert restarted >= B                                          # every world left at least once


@pytest.mark.gpu
def test_reset_boxes_a_restricted_spawn_area_restarts_inside_the_launch():
    """`device_reset = True` + `reset_boxes(world)`: a scenario whose agents start in [-0.25, 0.25)^2 and whose posts start in the
    right half of the arena -- its reset_world is `world.reset_boxes(...)`, and the program's in-launch restarts, its rollouts'
    resets and mpe_reset_rows all make the same draws."""
    boxes = [(-0.25, 0.25, -0.25, 0.25)] * 3 + [(0.2, 0.9, -0.9, 0.9)] * 3

    class Yard(Corral):
        device_reset = True
        arena = 0.95

        def reset_boxes(self, world):
            return boxes

        def reset_world(self, world, mask=None, seeds=None):
            idx = world.reset_boxes(boxes, mask, choices=[3], seeds=seeds)
            if world.choice_i32 is not None:
                world.choice_i32[0].copy_(World.merge_choice(world.choice_i32[0].long(), idx[:, 0].to(world.device), mask).to(torch.int32))

    def make(B, **kw):
        sc = Yard()
        w = sc.make_world(batch_size=B)
        w.seed = 21
        sc.reset_world(w)
        env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, compile_program=False, **kw)
        env.scenario = sc
        return env
    B = 4096
    env = make(B, max_episode_steps=1000, auto_reset=True)
    assert env._prog.struct.reset_boxes == 1 and env._device_restart_ok and env._episode_in_launch
    env.reset()
    p = env.world.pos
    assert float(p[:3].abs().max()) < 0.25 and float(p[3:, 0].min()) >= 0.2 and float(p[3:, 0].max()) < 0.9 and float(p[3:, 1].abs().max()) < 0.9
    act = torch.zeros((3, B, 5), device="cuda")
    act[:, :, 1] = 1.0
    restarted = 0
    for t in range(40):
        obs, rew, done, _ = env.step(act)
        fin = done[0] | done[1] | done[2]
        if bool(fin.any()):
            restarted += int(fin.sum())
            assert float(env.world.pos[:3][:, :, fin].abs().max()) < 0.25 and float(env.world.pos[3:, 0][:, fin].min()) >= 0.2
            assert bool((env.episode_step[fin] == 0).all())
    assert restarted >= B
    # the host-side seeded form places in the same boxes
    env.reset(seeds=list(range(B)))
    assert float(env.world.pos[:3].abs().max()) < 0.25 and float(env.world.pos[3:, 0].min()) >= 0.2
    # rollouts: the fused launch (resets drawn in the kernel) == per-step launches with mpe_reset_rows at the boundaries, to the bit
    from multiagent_particle_envs_amd.rollout import RandomRollout
    a, b = make(2048), make(2048)
    ra, rb = RandomRollout(a, episode_len=4, pool=4, regenerate=True), RandomRollout(b, episode_len=4, pool=4, regenerate=True)
    ra.enqueue(11)
    rb.fused(11)
    torch.cuda.synchronize()
    assert torch.equal(a.world.pos, b.world.pos) and torch.equal(a.world.vel, b.world.vel) and torch.equal(a.world.choice_i32, b.world.choice_i32)
    assert torch.equal(a._sets[(11 - 1) & 1].obs, b._sets[0].obs)
