"""symtrace.py: reference-style Scenario files traced into expression graphs -- CPU only (the device side: tests/test_gpu_traced.py).

  * the tracer itself on small scenarios written here: forks on symbolic conditions merged into selects, picks as proxies and
    enumerated where Python needs them concretely, what is refused and why, and that np.random / the file's builtins are left as
    they were;
  * the fixture files of tests/refstyle/ and -- in the build container -- the reference's nine files, loaded by path, unmodified:
    the trace reproduces the file's own reset_world / observation / reward on random worlds (fp64, to the last bit);
  * the COMMITTED traces of the nine (tests/golden/traced_*.json, tests/golden/gen_traced.py) evaluated with NumPy on the states
    the reference's own env recorded (tests/golden/*.npz) against the rows and rewards it recorded -- no reference tree needed;
  * the generated device code compiles (hipcc --genco needs no GPU).
"""
import json
import math
import os

import numpy as np
import pytest
import torch

import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import compat, refstyle, symtrace

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = os.path.join(HERE, "refstyle")
GOLDEN = os.path.join(HERE, "golden")
REF_SCENARIOS = "/root/reference/multiagent/scenarios"
NINE = ["simple", "simple_spread", "simple_tag", "simple_adversary", "simple_push", "simple_speaker_listener", "simple_reference",
        "simple_crypto", "simple_world_comm"]

compat.install()
from multiagent.core import World, Agent, Landmark  # noqa: E402
from multiagent.scenario import BaseScenario  # noqa: E402


class _Base(BaseScenario):
    """Two agents, two landmarks, everything default; subclasses override what a test is about."""

    def make_world(self):
        world = World()
        world.dim_c = 2
        world.agents = [Agent() for _ in range(2)]
        for i, a in enumerate(world.agents):
            a.name, a.silent, a.size = "agent %d" % i, i == 0, 0.1
        world.landmarks = [Landmark() for _ in range(2)]
        for i, l in enumerate(world.landmarks):
            l.name, l.movable, l.collide = "landmark %d" % i, False, False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for a in world.agents:
            a.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            a.state.p_vel = np.zeros(world.dim_p)
            a.state.c = np.zeros(world.dim_c)
        for l in world.landmarks:
            l.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            l.state.p_vel = np.zeros(world.dim_p)

    def reward(self, agent, world):
        return -np.sum(np.square(agent.state.p_pos - world.landmarks[0].state.p_pos))

    def observation(self, agent, world):
        return np.concatenate([agent.state.p_vel] + [l.state.p_pos - agent.state.p_pos for l in world.landmarks])


def test_forks_on_symbolic_conditions_merge_into_selects():
    class S(_Base):
        def reward(self, agent, world):
            rew = 0
            for l in world.landmarks:
                d = np.sqrt(np.sum(np.square(agent.state.p_pos - l.state.p_pos)))
                if d < 0.5:                       # a Python `if` on the state
                    rew -= 1
                    if d < 0.2:                   # ... nested
                        rew -= d * 10
                elif abs(agent.state.p_pos[0]) > 0.9 or not (agent.state.p_pos[1] < 0.9):
                    rew += 0.25
            return rew + min(agent.state.p_vel[0], 0.3) + max([agent.state.p_vel[1], -0.1, 0.0])
    sc = S()
    t = symtrace.trace(sc)
    assert t.paths["rew"][0] > 4 and t.paths["obs"] == [1, 1]            # control flow was explored; min / max did not fork
    assert symtrace.verify(sc, t, worlds=256) == 0.0
    ops = set(n.op for n in symtrace.topo(t.rew))
    assert "ite" in ops and "min" in ops and "max" in ops and "lt" in ops


def test_a_condition_asked_twice_forks_once():
    class S(_Base):
        def near(self, a, b):
            return True if np.sqrt(np.sum(np.square(a.state.p_pos - b.state.p_pos))) < 0.3 else False

        def observation(self, agent, world):
            flags = [np.array([1.0]) if self.near(agent, l) else np.array([-1.0]) for l in world.landmarks]
            again = [np.array([2.0]) if self.near(agent, l) else np.array([0.0]) for l in world.landmarks]
            return np.concatenate([agent.state.p_pos] + flags + again)
    sc = S()
    t = symtrace.trace(sc)
    assert t.paths["obs"] == [4, 4] and symtrace.verify(sc, t) == 0.0


def test_picks_are_proxies_and_are_enumerated_where_python_needs_them_concretely():
    class S(_Base):
        def reset_world(self, world):
            _Base.reset_world(self, world)
            for i, l in enumerate(world.landmarks):
                l.color = np.array([0.1, 0.1, 0.1])
                l.index = i
            goal = np.random.choice(world.landmarks)         # an object chosen per world
            goal.color = np.array([0.9, 0.1, 0.1])           # a write through the choice
            for a in world.agents:
                a.goal = goal
                a.tint = np.zeros(3)
                a.tint[goal.index + 1] = 1.0                 # the choice as an array index: one trace per value
            world.flag = int(np.random.randint(0, 3))        # a number chosen per world, used as a Python int

        def reward(self, agent, world):
            return -np.sqrt(np.sum(np.square(agent.state.p_pos - agent.goal.state.p_pos))) - 0.1 * world.flag

        def observation(self, agent, world):
            return np.concatenate([agent.goal.state.p_pos - agent.state.p_pos, agent.goal.color, agent.tint] +
                                  [l.color for l in world.landmarks] + [world.agents[1].state.c])
    sc = S()
    t = symtrace.trace(sc)
    assert t.pops == [2, 3] and sorted(t.enumerated) == [0, 1]
    assert [d[0] for d in t.draws].count("choice") == 2
    assert symtrace.verify(sc, t, worlds=200) == 0.0
    assert "sel" in set(n.op for row in t.obs for n in symtrace.topo(row))
    # the same as data
    t2 = symtrace.from_dict(json.loads(json.dumps(symtrace.to_dict(t))))
    assert symtrace.hip_source(t2) == symtrace.hip_source(t) and symtrace.verify(sc, t2) == 0.0


def test_what_the_tracer_refuses_and_why():
    class Hidden(_Base):          # state the trace cannot see: every call answers differently
        calls = 0

        def reward(self, agent, world):
            Hidden.calls += 1
            return float(Hidden.calls)

    class Draws(_Base):
        def observation(self, agent, world):
            return np.concatenate([agent.state.p_pos, np.random.uniform(-1, 1, 1)])

    class Scripted(_Base):
        def make_world(self):
            w = _Base.make_world(self)
            w.agents[1].action_callback = lambda a, wo: None
            return w

    class ReadsAction(_Base):
        def reward(self, agent, world):
            return -np.sum(np.square(agent.action.u))

    class Unstored(_Base):        # random numbers reset_world drew and kept outside the state: up to 4 travel as per-world parameters
        def reset_world(self, world):
            _Base.reset_world(self, world)
            world.bonus = np.random.uniform(0, 1, 5)

        def reward(self, agent, world):
            return float(np.sum(world.bonus))

    class Explodes(_Base):
        def reward(self, agent, world):
            rew = 0
            for k in range(40):
                if agent.state.p_pos[0] * (k + 1) < 0.01 * k:
                    rew += 1
            return rew

    class Concretises(_Base):
        def reward(self, agent, world):
            return float("%.3f" % agent.state.p_pos[0])       # (float(x) alone is traced: the file sees a `float` that lets x through)

    class RandomSizes(_Base):     # per-world randomness in make_world: the B worlds of a batch share their physics constants
        def make_world(self):
            w = _Base.make_world(self)
            w.agents[0].size = float(np.random.uniform(0.05, 0.2))
            return w

    for cls, why in ((RandomSizes, "different entity counts or physics constants"), (Draws, "draws random numbers"), (Scripted, "scripted agents"), (ReadsAction, "reads agent.action.u"),
                     (Unstored, "kept outside the state"), (Explodes, "control-flow paths"), (Concretises, "Python float")):
        with pytest.raises(symtrace.TraceUnsupported, match=why):
            symtrace.trace(cls(), max_paths=512 if cls is Explodes else None)
    sc = Hidden()
    with pytest.raises(symtrace.TraceUnsupported, match="does not reproduce"):
        symtrace.verify(sc, symtrace.trace(sc))


def test_tracing_leaves_numpy_random_and_the_files_namespace_as_they_were():
    sc = mpe.scenarios.load(os.path.join(FIXTURES, "herd.py")).Scenario()
    space = type(sc).reward.__globals__
    np.random.seed(123)
    before = np.random.get_state()[1].copy()
    fns = (np.random.uniform, np.random.choice, np.random.randint, np.random.randn)
    t = symtrace.trace(sc)
    symtrace.verify(sc, t)
    assert (np.random.uniform, np.random.choice, np.random.randint, np.random.randn) == fns
    assert np.array_equal(np.random.get_state()[1], before)                   # the caller's stream was not consumed
    assert "min" not in space and "any" not in space                          # the injected builtins are gone again
    with pytest.raises(symtrace.TraceUnsupported):                            # ... also after a refused trace
        symtrace.trace(mpe.scenarios.load(os.path.join(FIXTURES, "patrol.py")).Scenario())
    assert (np.random.uniform, np.random.choice) == fns[:2]


@pytest.mark.parametrize("name", ["herd", "relay", "convoy", "survey", "mesh"])
def test_fixture_files_trace_and_reproduce_their_own_callbacks(name):
    sc = mpe.scenarios.load(os.path.join(FIXTURES, name + ".py")).Scenario()
    ts = refstyle.trace_ref_scenario(sc, cache=False)
    ulp = 1e-15 if name in ("mesh", "survey") else 0.0          # (np.linalg.norm sums in its own order -- BLAS dot, pairwise: one ulp)
    assert ts.t.verified <= ulp
    assert symtrace.verify(sc, ts.t, worlds=300, seed=7) <= ulp
    # every coordinate a uniform draw of its own: restarts can be drawn on the device -- relay / convoy as World.reset_uniform's
    # placement, herd (agents on [-0.8, 0.8)^2) in per-entity boxes
    assert ts.device_reset and ts.landmark_range == {"herd": 1.0, "survey": 0.8, "mesh": 1.0}.get(name, 0.9)
    assert (ts.reset_boxes(None) is None) == (name != "herd")
    if name == "herd":
        assert ts.reset_boxes(None)[0] == (-0.8, 0.8, -0.8, 0.8) and ts.reset_boxes(None)[3] == (-1.0, 1.0, -1.0, 1.0)


def test_a_reset_world_that_is_not_a_formula_of_its_draws_runs_on_the_host_and_the_callbacks_are_still_traced():
    """tests/refstyle/scatter.py places entities by rejection sampling (`while` the spot is taken: draw again) and draws normal
    velocities: tracing reset_world would never end / is refused.  The trace keeps observation and reward (verified as always),
    follows `world.goal = np.random.choice(world.landmarks)` as a per-world pick, and marks reset_world as the file's own Python, run per restarting world with np.random seeded per world -- a seeded reset
    is the reference's `np.random.seed(s); env.reset()` value for value."""
    sc = mpe.scenarios.load(os.path.join(FIXTURES, "scatter.py")).Scenario()
    np.random.seed(11)
    before = np.random.get_state()[1].copy()
    ts = refstyle.trace_ref_scenario(sc, cache=False)
    t = ts.t
    assert np.array_equal(np.random.get_state()[1], before)
    assert t.host_reset and "state-dependent decisions" in t.host_reset and t.n_u == 0
    assert t.pops == [3] and t.draws == [("choice", 3)]          # world.goal = np.random.choice(world.landmarks): still followed as a pick
    assert t.verified <= 1e-15 and symtrace.verify(sc, t, worlds=150, seed=3) <= 1e-15
    assert not ts.device_reset and ts.reset_boxes(None) is None and "runs on the host" in ts.report()
    B = 16
    w = ts.make_world(B, "cpu")
    w.seed, w.rng_mode = 0, "device"
    seeds = list(range(900, 900 + B))
    ts.reset_world(w, None, seeds)
    P, V = w.get_state(all_entities=True)
    cw = sc.make_world()
    for b, s in enumerate(seeds):
        np.random.seed(s)
        sc.reset_world(cw)
        ents = cw.agents + cw.landmarks
        assert np.abs(P[b] - np.array([e.state.p_pos for e in ents])).max() < 1e-6
        assert np.abs(V[b] - np.array([e.state.p_vel for e in ents])).max() < 1e-6
        assert int(w.choice_i32[0][b]) == cw.landmarks.index(cw.goal)          # ... and the pick of that world is the file's
    gaps = np.linalg.norm(P[:, :, None, :] - P[:, None, :, :], axis=-1) + 10.0 * np.eye(P.shape[1])
    assert gaps.min() >= 0.3 - 1e-6                                   # the file's own invariant: nobody spawns on anybody
    # a masked restart touches only the masked worlds; without seeds the stream is (env seed, episode, world): reproducible, and the
    # caller's np.random stream is left alone
    mask = torch.zeros(B, dtype=torch.bool)
    mask[3] = mask[7] = True
    np.random.seed(11)
    ts.reset_world(w, mask, None)
    assert np.array_equal(np.random.get_state()[1], before)
    P2, _ = w.get_state(all_entities=True)
    changed = np.abs(P2 - P).max(axis=(1, 2)) > 0
    assert changed.tolist() == mask.tolist()
    # as data: the flag travels; a trace without its file cannot reset
    t2 = symtrace.from_dict(json.loads(json.dumps(symtrace.to_dict(t))))
    assert t2.host_reset == t.host_reset
    with pytest.raises(RuntimeError, match="needs the scenario file"):
        refstyle.TracedRefScenario(None, t2).reset_world(w, None, None)


def test_a_loop_on_a_state_dependent_condition_in_a_callback_is_refused_not_run_forever():
    class Halving(_Base):
        def reward(self, agent, world):
            d = abs(agent.state.p_pos[0]) + 1.0
            while d > 0.1:
                d = d * 0.5
            return d
    with pytest.raises(symtrace.TraceUnsupported, match="state-dependent decisions on one path"):
        symtrace.trace(Halving())


def test_patrol_stays_on_the_host_path_with_the_reason():
    sc = mpe.scenarios.load(os.path.join(FIXTURES, "patrol.py")).Scenario()
    with pytest.raises(symtrace.TraceUnsupported, match="scripted agents"):
        refstyle.trace_ref_scenario(sc, cache=False)
    env = mpe.make_env(os.path.join(FIXTURES, "patrol.py"), batch_size=2, device="cpu")
    assert not env.traced and "scripted agents" in env.trace_fallback
    with pytest.raises(symtrace.TraceUnsupported):
        mpe.make_env(os.path.join(FIXTURES, "patrol.py"), batch_size=2, device="cpu", traced=True)
    # a CPU world never takes the traced path (its program runs compiled in on a HIP device): the host path, with the reason
    env = mpe.make_env(os.path.join(FIXTURES, "herd.py"), batch_size=2, device="cpu")
    assert not env.traced and "HIP device" in env.trace_fallback


@pytest.mark.skipif(not os.path.isdir(REF_SCENARIOS), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("name", NINE)
def test_the_nine_reference_files_trace_and_reproduce_their_own_callbacks(name):
    """The file as shipped, loaded by path: reset_world, every agent's observation and reward as graphs == the file's own NumPy
    code on random worlds, bit for bit in fp64; and the committed trace (what the GPU box runs) is this trace."""
    path = os.path.join(REF_SCENARIOS, name + ".py")
    before = open(path, "rb").read()
    sc = mpe.scenarios.load(path).Scenario()
    try:          # (as tests/golden/gen_traced.py: with benchmark_data where the file's works)
        t = symtrace.trace(sc, want_info=hasattr(sc, "benchmark_data"))
    except symtrace.TraceUnsupported:
        t = symtrace.trace(sc)
    assert (t.info is not None) == (name in ("simple_spread", "simple_tag", "simple_adversary", "simple_crypto", "simple_world_comm"))
    assert symtrace.verify(sc, t, worlds=200, seed=3) == 0.0
    assert open(path, "rb").read() == before
    with open(os.path.join(GOLDEN, "traced_%s.json" % name)) as fh:
        committed = symtrace.from_dict(json.load(fh))
    assert symtrace.hip_source(committed) == symtrace.hip_source(t), "tests/golden/traced_%s.json is stale: python tests/golden/gen_traced.py" % name
    assert symtrace.verify(sc, committed, worlds=50, seed=4) == 0.0


def _golden_states(g, t_, tr):
    """(P [W, E, 2], V [W, E, 2], Cw [W, A, dim_c], K [W, picks]) of recorded step t_ (None: the reset state)."""
    pos = g["pos0"] if t_ is None else g["pos"][t_]
    vel = g["vel0"] if t_ is None else g["vel"][t_]
    W = pos.shape[0]
    V = np.zeros((W, tr.E, 2))
    V[:, :vel.shape[1]] = vel
    Cw = np.zeros((W, tr.A, tr.dim_c))
    if t_ is not None and tr.dim_c:
        for i in range(tr.A):
            if "c%d" % i in g:
                Cw[:, i] = g["c%d" % i][t_][:, :tr.dim_c]
    K = g["choice"].astype(np.int64) if "choice" in g and g["choice"].shape[1] else np.zeros((W, len(tr.pops)), np.int64)
    return pos.astype(np.float64), V, Cw, K


@pytest.mark.parametrize("name", NINE)
def test_committed_traces_of_the_nine_against_the_reference_goldens(name, golden):
    """No reference tree, no GPU: the committed graphs evaluated (NumPy, fp64) on the states the reference's own env recorded give
    the rows and rewards it recorded -- per step, per world, per agent -- and its seeded resets."""
    with open(os.path.join(GOLDEN, "traced_%s.json" % name)) as fh:
        tr = symtrace.from_dict(json.load(fh))
    g = golden(name if name in ("simple", "simple_spread", "simple_tag") else "f3_" + name)
    T, W, A = g["rew"].shape
    assert A == tr.A
    worst = 0.0
    shared = bool(tr.collaborative)
    for t_ in [None] + list(range(T)):
        P, V, Cw, K = _golden_states(g, t_, tr)
        roots = [n for row in tr.obs for n in row] + list(tr.rew)
        vals = symtrace.evaluate(roots, W, P=P, V=V, Cw=Cw, K=K)
        off = np.cumsum([0] + [len(r) for r in tr.obs])
        for i in range(A):
            want = g["obs_reset%d" % i] if t_ is None else g["obs%d" % i][t_]
            got = np.stack(vals[off[i]:off[i + 1]], axis=1)
            worst = max(worst, float(np.abs(got - want).max()))
        if t_ is not None and tr.info is not None and name in ("simple_spread", "simple_tag", "simple_world_comm"):
            # benchmark_data as the reference's env recorded it (ints exactly)
            keys = ("info_rew", "info_collisions", "info_min_dists", "info_occupied") if name == "simple_spread" else ("info_collisions",)
            for i in range(A):
                iv = symtrace.evaluate(tr.info[i], W, P=P, V=V, Cw=Cw, K=K)
                for k, key in enumerate(keys):
                    if "collisions" in key or "occupied" in key:
                        assert np.array_equal(iv[k], g[key][t_][:, i]), (key, t_, i)
                    else:
                        assert np.abs(iv[k] - g[key][t_][:, i]).max() <= 1e-9
        if t_ is not None:
            rew = np.stack(vals[off[-1]:off[-1] + A], axis=1)               # [W, A]
            if shared:                                            # environment.py:100-102
                rew = np.repeat(rew.sum(axis=1, keepdims=True), A, axis=1)
            worst = max(worst, float((np.abs(rew - g["rew"][t_]) / np.maximum(1.0, np.abs(g["rew"][t_]))).max()))
    assert worst <= 1e-9, worst
    # the seeded reset: the file's own random stream replayed per world (np.random.seed(s); env.reset())
    ts = refstyle.TracedRefScenario(None, tr)
    U = np.zeros((W, max(tr.n_u, 1)))
    K = np.zeros((W, len(tr.pops)), np.int64)
    for b in range(W):
        rs, iu, ik = np.random.RandomState(int(g["seeds"][b])), 0, 0
        for d in tr.draws:
            if d[0] == "uniform":
                U[b, iu:iu + d[3]] = rs.random_sample(d[3])
                iu += d[3]
            else:
                K[b, ik] = rs.randint(0, d[1])
                ik += 1
    pos = np.stack(symtrace.evaluate([n for e in tr.reset_pos for n in e], W, K=K, U=U), axis=1).reshape(W, tr.E, 2)
    same = np.all(np.abs(pos - g["pos0"]) <= 1e-12, axis=(1, 2))      # (the recorders squeeze / stage some worlds after the reset)
    if "staged" in g:
        assert same[~g["staged"]].all() and (~g["staged"]).sum() >= W // 8
    else:
        assert same.sum() >= W // 2
    if "choice" in g and g["choice"].shape[1]:
        assert np.array_equal(K, g["choice"])
    assert ts.device_reset                                            # the nine all place uniformly: device-side restarts apply


def test_generated_device_code_compiles():
    """symtrace.hip_source appended to the generated header of the program: hipcc --genco accepts it (no GPU needed)."""
    from multiagent_particle_envs_amd import _build
    sc = mpe.scenarios.load(os.path.join(FIXTURES, "relay.py")).Scenario()
    ts = refstyle.trace_ref_scenario(sc)
    w = ts.make_world(4, "cpu")
    env = mpe.MultiAgentEnv(w, ts.reset_world, None, None, compile_program=False)
    assert env._prog is not None and env._prog.traced and env._prog.struct.traced == 1 and env.fused
    src = env._prog.static_source(env._desc)
    assert "MPE_ROWS_TRACED" in src and "traced_obs" in src and "sqrt_lt" not in src.split("traced_obs")[0]
    assert len(_build.compile_rows_image(src)) > 10000
    # a program with code ops but no source is refused on the host already
    from multiagent_particle_envs_amd import rowspec, _abi
    with pytest.raises(_abi.MpeError, match="needs the source"):
        rowspec.RowProgram(w, [ts.obs_spec(a, w) for a in w.agents], [ts.reward_spec(a, w) for a in w.agents])


def test_the_torch_evaluator_of_a_reset_program_agrees_with_the_numpy_one():
    """A reset_world that is not World.reset_uniform's placement (herd: agents on [-0.8, 0.8)^2) is drawn and evaluated with torch
    ops on the device (symtrace.evaluate_torch): the same graph, the same numbers as the NumPy evaluation (fp32 vs fp64)."""
    import torch
    sc = mpe.scenarios.load(os.path.join(FIXTURES, "herd.py")).Scenario()
    t = symtrace.trace(sc)
    B = 300
    rs = np.random.RandomState(0)
    U = rs.rand(B, t.n_u)
    K = np.stack([rs.randint(0, n, B) for n in t.pops], axis=1)
    flat = [n for e in t.reset_pos for n in e] + [n for e in t.reset_vel for n in e]
    a = np.stack(symtrace.evaluate(flat, B, K=K, U=U))
    b = torch.stack(symtrace.evaluate_torch(flat, B, K=torch.as_tensor(K).t(), U=torch.as_tensor(U, dtype=torch.float32).t(), device="cpu"))
    assert np.abs(a - b.numpy()).max() <= 2e-7 and np.abs(a[:2 * t.A]).max() < 0.8 and np.abs(a[:2 * t.A]).max() > 0.75
    # graphs with picks, selections and comparisons too (observation / reward graphs of a scenario with picks)
    sc = mpe.scenarios.load(os.path.join(FIXTURES, "convoy.py")).Scenario()
    t = symtrace.trace(sc)
    P, V, Cw = symtrace.random_states(t, B, rs)
    K = np.stack([rs.randint(0, n, B) for n in t.pops], axis=1)
    roots = [n for row in t.obs for n in row] + list(t.rew)
    a = np.stack(symtrace.evaluate(roots, B, P=P.astype(np.float32).astype(np.float64), V=V.astype(np.float32).astype(np.float64), Cw=Cw, K=K))
    tp = torch.as_tensor(P, dtype=torch.float32).permute(1, 2, 0)
    tv = torch.as_tensor(V, dtype=torch.float32).permute(1, 2, 0)
    b = torch.stack([x.to(torch.float32) for x in symtrace.evaluate_torch(roots, B, K=torch.as_tensor(K).t(), P=tp, V=tv, device="cpu")]).numpy()
    ok = symtrace.decision_margin(roots, B, P=P, V=V, Cw=Cw, K=K) > 1e-5
    assert ok.mean() > 0.9 and (np.abs(a - b)[:, ok] / np.maximum(1.0, np.abs(a[:, ok]))).max() <= 1e-5


# ---- predication: value-only control flow without forks (symtrace.predicated_twin) --------------------------------------------------
_PREDICATION_FILE = '''
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.dim_c = 2
        world.agents = [Agent() for _ in range(N_AGENTS)]
        for i, a in enumerate(world.agents):
            a.name, a.silent, a.size, a.lead = "agent %d" % i, True, 0.08, i == 0
        world.landmarks = [Landmark() for _ in range(2)]
        for l in world.landmarks:
            l.movable, l.collide, l.size = False, False, 0.1
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for e in world.agents + world.landmarks:
            e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            e.state.p_vel = np.zeros(world.dim_p)
        for a in world.agents:
            a.state.c = np.zeros(world.dim_c)

    def touching(self, a, b):
        return True if np.sqrt(np.sum(np.square(a.state.p_pos - b.state.p_pos))) < a.size + b.size else False      # a conditional expression

    def band(self, x):                      # early returns
        if x < 0.2:
            return 0
        if x < 0.5:
            return (x - 0.2) * 3
        return min(np.exp(x - 0.5), 2.0)

    def reward(self, agent, world):
        rew = 0
        for other in world.agents:          # an `if` per other agent that only assigns: 2^(N-1) paths when it forks
            if other is not agent and self.touching(other, agent):
                rew -= 1
        x = abs(agent.state.p_pos[0])
        if x > 0.9:                         # if / elif / else of assignments; `bonus` exists on every branch
            bonus = -x
        elif x > 0.5 and not agent.lead:
            bonus = 0.25
        else:
            bonus = 0.0
        if agent.state.p_pos[1] > 0.3:      # a name only this branch creates: merged (it keeps its value on the other side)
            extra = 1.0
            rew += extra
        label = None
        if agent.state.p_pos[0] > 0.95:     # a string or None: nothing to select between, this `if` forks
            label = "edge"
        return rew + bonus - self.band(abs(agent.state.p_pos[1])) + (0.125 if label else 0.0)

    def observation(self, agent, world):
        seen, flags = [], [np.array([-1.0]), np.array([-1.0])]
        for k, l in enumerate(world.landmarks):
            if self.touching(agent, l) or agent.lead:       # branches that APPEND, and an element assignment
                seen.append(l.state.p_pos - agent.state.p_pos)
                flags[k] = np.array([1.0])
            else:
                seen.append([0, 0])
        return np.concatenate([agent.state.p_vel, agent.state.p_pos] + seen + flags)
'''


@pytest.mark.parametrize("n_agents", [3, 14])
def test_value_only_control_flow_is_predicated_not_forked(tmp_path, n_agents):
    path = tmp_path / ("predication_%d.py" % n_agents)
    path.write_text(_PREDICATION_FILE.replace("N_AGENTS", str(n_agents)))
    sc = mpe.scenarios.load(str(path)).Scenario()
    t = symtrace.trace(sc)
    assert t.predicated and t.paths["obs"] == [1] * n_agents and t.paths["rew"] == [2] * n_agents      # (the one `if` that cannot merge)
    assert symtrace.verify(sc, t, worlds=300) == 0.0
    if n_agents == 3:          # the forking trace of the same file: more paths, the same function
        f = symtrace.trace(sc, predicate=False)
        assert not f.predicated and f.paths["rew"][0] > 20 and f.paths["obs"][1] == 4 and symtrace.verify(sc, f, worlds=300) == 0.0
    else:                      # 13 other agents: 2^13 paths per agent for the forking tracer -- refused; predicated: instant
        with pytest.raises(symtrace.TraceUnsupported, match="control-flow paths"):
            symtrace.trace(sc, predicate=False)


_VECTOR_FILE = '''
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    """Written by somebody who thinks in arrays: distance matrices, comparisons of whole arrays, reductions along an axis."""
    def make_world(self):
        world = World()
        world.agents = [Agent() for _ in range(N_AGENTS)]
        for i, a in enumerate(world.agents):
            a.name, a.silent, a.size = "agent %d" % i, True, 0.08
        world.landmarks = [Landmark() for _ in range(N_AGENTS)]
        for l in world.landmarks:
            l.movable, l.collide, l.size = False, False, 0.1
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for e in world.agents + world.landmarks:
            e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            e.state.p_vel = np.zeros(world.dim_p)
        for a in world.agents:
            a.state.c = np.zeros(world.dim_c)

    def zone(self, p):                      # nested early returns
        if abs(p[0]) < 0.5:
            if abs(p[1]) < 0.5:
                return 2.0
            return 1.0
        elif abs(p[0]) < 0.8:
            return 0.5
        else:
            return 0.0

    def reward(self, agent, world):
        X = np.array([a.state.p_pos for a in world.agents])
        Y = np.array([l.state.p_pos for l in world.landmarks])
        D = np.sqrt(((X[:, None, :] - Y[None, :, :]) ** 2).sum(-1))          # [agents, landmarks]
        rew = -D.min(axis=0).sum()                                           # the nearest agent of every landmark
        rew += 0.1 * (D < 0.2).sum() + 0.05 * np.count_nonzero(D.min(axis=1) < 0.1)
        if (D.min(axis=0) < 0.15).all():
            rew += 5.0
        if np.any(np.abs(X) > 0.95):
            rew -= 1.0
        for a in world.agents:
            if a is agent:
                continue
            gap = np.linalg.norm(a.state.p_pos - agent.state.p_pos)
            if gap > 0.5:
                continue
            rew -= 0.5 - gap
        if -0.25 < agent.state.p_pos[0] < 0.25:
            rew += 0.01
        return rew + self.zone(agent.state.p_pos) * 0.01

    def observation(self, agent, world):
        Y = np.array([l.state.p_pos for l in world.landmarks]) - agent.state.p_pos
        near = np.linalg.norm(Y, axis=1) < 0.6
        return np.concatenate([agent.state.p_vel, agent.state.p_pos, (Y * near[:, None]).reshape(-1), Y.max(axis=0), np.clip(Y, -0.5, 0.5).min(axis=0)])
'''


@pytest.mark.parametrize("n_agents", [3, 8])
def test_array_comparisons_axis_reductions_continue_and_nested_returns_do_not_fork(tmp_path, n_agents):
    """`D < 0.2` on an array (NumPy's own `<` stores bools: one fork per element), `D.min(axis=0)` / `.all()` / `.any()` as
    METHODS, `if T: continue`, nested early returns, `a < x < b`: predicated in the twin -- one path whatever the team size; the
    forking trace of the same file is refused already at N = 3 (more than 8192 paths)."""
    path = tmp_path / ("vector_%d.py" % n_agents)
    path.write_text(_VECTOR_FILE.replace("N_AGENTS", str(n_agents)))
    sc = mpe.scenarios.load(str(path)).Scenario()
    t = symtrace.trace(sc)
    assert t.predicated and t.paths["obs"] == [1] * n_agents and t.paths["rew"] == [1] * n_agents
    assert symtrace.verify(sc, t, worlds=200) <= 1e-15
    with pytest.raises(symtrace.TraceUnsupported, match="control-flow paths"):      # (2^9 for `D < 0.2` alone at N = 3)
        symtrace.trace(sc, predicate=False)
    if n_agents == 8:
        ts = refstyle.trace_ref_scenario(sc, cache=False)          # ... and it fits the kernel: the shared reward once per world
        assert ts.t.n_shared >= 2 and len(ts.row_source(None).splitlines()) < 6000


_NEAREST_FILE = '''
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.agents = [Agent() for _ in range(3)]
        for i, a in enumerate(world.agents):
            a.name, a.silent = "agent %d" % i, True
        world.landmarks = [Landmark() for _ in range(7)]
        for l in world.landmarks:
            l.movable, l.collide = False, False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for e in world.agents + world.landmarks:
            e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            e.state.p_vel = np.zeros(world.dim_p)
        for a in world.agents:
            a.state.c = np.zeros(world.dim_c)

    def reward(self, agent, world):
        target = min(world.landmarks, key=lambda l: np.sum(np.square(l.state.p_pos - agent.state.p_pos)))      # the OBJECT
        return -np.linalg.norm(target.state.p_pos - agent.state.p_pos)

    def observation(self, agent, world):
        d = [np.linalg.norm(l.state.p_pos - agent.state.p_pos) for l in world.landmarks]
        k = int(np.argmin(d))                                                   # an index
        ranked = sorted(d)                                                      # values in order: no decision at all
        others = [np.linalg.norm(a.state.p_pos - agent.state.p_pos) for a in world.agents if a is not agent]
        others.sort(reverse=True)
        return np.concatenate([agent.state.p_pos, world.landmarks[k].state.p_pos - agent.state.p_pos, ranked[:3], np.sort(np.array(d))[-2:], others])
'''


_COUNTING_FILE = '''
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.agents = [Agent() for _ in range(5)]
        for i, a in enumerate(world.agents):
            a.name, a.silent, a.size = "agent %d" % i, True, 0.1
        world.landmarks = [Landmark() for _ in range(6)]
        for l in world.landmarks:
            l.movable, l.collide = False, False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for e in world.agents + world.landmarks:
            e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            e.state.p_vel = np.zeros(world.dim_p)
        for a in world.agents:
            a.state.c = np.zeros(world.dim_c)

    def near(self, a, b, r):
        return np.linalg.norm(a.state.p_pos - b.state.p_pos) < r

    def reward(self, agent, world):
        bumps = len([a for a in world.agents if a is not agent and self.near(a, agent, 0.2)])
        reach = sum(np.linalg.norm(l.state.p_pos - agent.state.p_pos) for l in world.landmarks if self.near(l, agent, 0.8))
        crowd = sum(1 for a in world.agents if self.near(a, agent, 0.5) if a is not agent)
        best = min(np.linalg.norm(l.state.p_pos - agent.state.p_pos) for l in world.landmarks if l.state.p_pos[0] > -5.0)
        north = 1.0 if any(l.state.p_pos[1] > 0 for l in world.landmarks if self.near(l, agent, 0.6)) else 0.0
        calm = 1.0 if all(abs(a.state.p_vel[0]) < 0.5 for a in world.agents if self.near(a, agent, 0.6)) else 0.0
        return -bumps - 0.1 * reach - 0.01 * crowd - best + north + calm

    def observation(self, agent, world):
        return np.concatenate([agent.state.p_vel, agent.state.p_pos] + [a.state.p_pos - agent.state.p_pos for a in world.agents if a is not agent])
'''


def test_counting_and_summing_over_filtered_comprehensions_does_not_fork(tmp_path):
    """`len([a for a in others if touching(a)])`, `sum(d(l) for l in landmarks if near(l))`, min / any / all over a filtered
    generator: how LONG the list is would be a decision per element (2^N); the twin moves the filter into the element (`E if C else
    0`), which selects.  A filter that is an ordinary truth value (`if a is not agent`) filters as written."""
    path = tmp_path / "counting.py"
    path.write_text(_COUNTING_FILE)
    sc = mpe.scenarios.load(str(path)).Scenario()
    t = symtrace.trace(sc)
    assert t.predicated and max(t.paths["obs"] + t.paths["rew"]) == 1 and [len(r) for r in t.obs] == [12] * 5
    assert symtrace.verify(sc, t, worlds=300) <= 1e-15
    with pytest.raises(symtrace.TraceUnsupported, match="control-flow paths"):
        symtrace.trace(sc, predicate=False)


def test_nearest_of_n_is_n_paths_and_sorted_values_need_no_decision(tmp_path):
    """np.argmin / `min(objects, key=...)`: which element is smallest is a decision with N outcomes -- N paths (one test per candidate),
    where the running comparison of NumPy / Python forks 2^(N-1) ways; the VALUES of sorted() / np.sort / list.sort() come out of a
    network of min / max pairs: no decision."""
    path = tmp_path / "nearest.py"
    path.write_text(_NEAREST_FILE)
    sc = mpe.scenarios.load(str(path)).Scenario()
    t = symtrace.trace(sc)
    assert t.predicated and t.paths["obs"] == [7] * 3 and t.paths["rew"] == [7] * 3
    assert symtrace.verify(sc, t, worlds=300) <= 1e-15
    f = symtrace.trace(sc, predicate=False)          # (without the twin: list.sort() is Python's own -- it forks; the functions do not)
    assert f.paths["rew"] == [7] * 3 and f.paths["obs"] == [14] * 3 and symtrace.verify(sc, f, worlds=100) <= 1e-15
    assert sorted([3, 1, 2]) == [1, 2, 3] and np.argmin([3, 1, 2]) == 1


def test_random_numbers_kept_outside_the_state_travel_as_per_world_parameters():
    """`world.goal_pos = np.random.uniform(-0.8, 0.8, 2)` -- a goal that is coordinates, not an entity -- read by the callbacks: each
    such draw occupies a pick slot (a "pick" among 2^24 values, read back as k * 2^-24), so seeded resets, device restarts and the
    kernel's pick accessor serve it like any pick."""
    class S(_Base):
        def reset_world(self, world):
            world.mark = np.random.choice(world.landmarks)
            world.goal_pos = np.random.uniform(-0.8, +0.8, world.dim_p)
            _Base.reset_world(self, world)
            world.pace = np.random.uniform(0.5, 1.5)

        def reward(self, agent, world):
            return -np.linalg.norm(agent.state.p_pos - world.goal_pos) * world.pace - 0.1 * np.linalg.norm(agent.state.p_pos - world.mark.state.p_pos)

        def observation(self, agent, world):
            return np.concatenate([agent.state.p_pos, world.goal_pos - agent.state.p_pos])
    sc = S()
    ts = refstyle.trace_ref_scenario(sc, cache=False)
    t = ts.t
    n_lm = len(sc.make_world().landmarks)
    assert t.real_picks() == 1 and t.pops == [n_lm] + [symtrace.PARAM_POP] * 3 and len(t.params) == 3 and t.verified <= 1e-15
    assert "U" not in symtrace.inputs_of([n for row in t.obs for n in row] + list(t.rew))          # the callbacks read pick slots
    assert ts.device_reset and "per-world parameters" in ts.report()                               # none of the three places an entity
    t2 = symtrace.from_dict(json.loads(json.dumps(symtrace.to_dict(t))))
    assert t2.params == t.params and t2.real_picks() == 1 and symtrace.hip_source(t2) == symtrace.hip_source(t)
    assert "K(1)" in symtrace.hip_source(t) and "K(3)" in symtrace.hip_source(t)
    # a seeded reset: the file's own stream; what the callbacks then see is the file's numbers to 24 bits
    B = 12
    w = ts.make_world(B, "cpu")
    w.seed, w.rng_mode = 0, "device"
    seeds = list(range(300, 300 + B))
    ts.reset_world(w, None, seeds)
    P, V = w.get_state(all_entities=True)
    K = w.choice_i32.cpu().numpy().T
    roots = [n for row in t.obs for n in row] + list(t.rew)
    vals = symtrace.evaluate(roots, B, P=P.astype(np.float64), V=V.astype(np.float64), Cw=np.zeros((B, t.A, t.dim_c)), K=K)
    cw = sc.make_world()
    for b, s in enumerate(seeds):
        np.random.seed(s)
        sc.reset_world(cw)
        assert K[b, 0] == cw.landmarks.index(cw.mark)
        for k, e in enumerate(cw.agents + cw.landmarks):
            assert np.abs(e.state.p_pos - P[b, k]).max() < 1e-6
            e.state.p_pos = P[b, k].astype(np.float64)
        want = np.concatenate([sc.observation(a, cw) for a in cw.agents] + [[sc.reward(a, cw)] for a in cw.agents])
        assert np.abs(np.array([v[b] for v in vals]) - want).max() < 5e-7
    # a draw that the callbacks read AND that places an entity: one source for both, so no device-side restart draw
    class Both(_Base):
        def reset_world(self, world):
            _Base.reset_world(self, world)
            world.spot = np.random.uniform(-1, +1, world.dim_p)
            world.landmarks[0].state.p_pos = world.spot

        def reward(self, agent, world):
            return -np.linalg.norm(agent.state.p_pos - world.spot)
    tb = refstyle.trace_ref_scenario(Both(), cache=False)
    assert len(tb.t.params) == 2 and not tb.device_reset


def test_rand_and_random_sample_are_the_draws_uniform_scales():
    """`np.random.rand(2) * 2 - 1`, `np.random.random()`: [0, 1) draws of the same stream uniform() takes its numbers from -- traced as
    U inputs, and a seeded reset replays them value for value (np.random.seed(s); reset_world(world) of the file itself)."""
    class S(_Base):
        def reset_world(self, world):
            for a in world.agents:
                a.state.p_pos = np.random.rand(world.dim_p) * 2.0 - 1.0
                a.state.p_vel = np.zeros(world.dim_p)
                a.state.c = np.zeros(world.dim_c)
            for l in world.landmarks:
                l.state.p_pos = np.array([np.random.random() - 0.5, np.random.uniform(-0.3, 0.3)]) + np.random.random_sample(2) * 0.1
                l.state.p_vel = np.zeros(world.dim_p)
    sc = S()
    t = symtrace.trace(sc)
    assert symtrace.verify(sc, t, worlds=100) <= 1e-15
    assert [d[0] for d in t.draws] == ["uniform"] * len(t.draws) and t.n_u == 2 * len(sc.make_world().agents) + 4 * len(sc.make_world().landmarks)
    world = sc.make_world()
    flat = [n for e in t.reset_pos for n in e]
    for seed in (3, 77):
        np.random.seed(seed)
        sc.reset_world(world)
        want = np.concatenate([e.state.p_pos for e in world.agents + world.landmarks])
        rs, U, iu = np.random.RandomState(seed), np.zeros((1, t.n_u)), 0
        for d in t.draws:          # (as refstyle.TracedRefScenario.reset_world replays a seed)
            U[0, iu:iu + d[3]] = rs.random_sample(d[3])
            iu += d[3]
        got = np.array([v[0] for v in symtrace.evaluate(flat, 1, U=U, K=np.zeros((1, 0), np.int64))])
        assert np.array_equal(got, want)
    # an affine function of ONE draw per coordinate, however it is written, is a box the device can draw restarts in; a coordinate
    # that mixes two draws (the landmarks here) is not
    ts = refstyle.TracedRefScenario(sc, t)
    assert ts._uniform_pattern() == (1.0, False, None)

    class Boxes(S):
        def reset_world(self, world):
            S.reset_world(self, world)
            for k, l in enumerate(world.landmarks):
                l.state.p_pos = np.array([0.5 - np.random.rand(), np.random.random() * 0.4 + 0.1 * k])
    sc = Boxes()
    t = symtrace.trace(sc)
    assert symtrace.verify(sc, t, worlds=50) <= 1e-15
    rng, dev, boxes = refstyle.TracedRefScenario(sc, t)._uniform_pattern()
    assert dev and boxes is not None and boxes[0] == (-1.0, 1.0, -1.0, 1.0)
    assert np.allclose(boxes[t.A], (-0.5, 0.5, 0.0, 0.4)) and np.allclose(boxes[t.A + 1], (-0.5, 0.5, 0.1, 0.5))


def test_a_team_too_large_for_straight_line_code_is_refused_with_the_reason(tmp_path, monkeypatch):
    """Every agent's functions spell out their whole graph: N^3 statements for a reward that visits every agent-landmark pair.
    Past MPE_TRACE_MAX_STATEMENTS the file stays on the host path (the trace itself is instant and exact)."""
    path = tmp_path / "big_team.py"
    path.write_text(_PREDICATION_FILE.replace("N_AGENTS", "12"))
    sc = mpe.scenarios.load(str(path)).Scenario()
    ts = refstyle.trace_ref_scenario(sc, cache=False)
    n = len(ts.row_source(None).splitlines())
    monkeypatch.setenv("MPE_TRACE_MAX_STATEMENTS", str(n // 2))
    with pytest.raises(symtrace.TraceUnsupported, match="statements of straight-line device code"):
        refstyle.trace_ref_scenario(sc, cache=False)
    env = mpe.make_env(str(path), batch_size=2, device="cpu")
    assert not env.traced and "straight-line" in env.trace_fallback


def test_the_fixtures_and_committed_traces_hardly_fork():
    for name in ("herd", "relay", "convoy", "survey", "mesh", "scatter"):
        sc = mpe.scenarios.load(os.path.join(FIXTURES, name + ".py")).Scenario()
        t = symtrace.trace(sc)
        assert t.predicated and max(t.paths["obs"] + t.paths["rew"]) == 1, (name, t.paths)
    for name in NINE:
        with open(os.path.join(GOLDEN, "traced_%s.json" % name)) as fh:
            paths = json.load(fh)["paths"]
        # (simple_crypto.py:104-113 -- `if (a.state.c == np.zeros(dim_c)).all(): continue / else:` -- forked 256 ways before array
        #  comparisons, `.all()` and `continue` were predicated)
        assert max(paths["obs"] + paths["rew"]) == 1, (name, paths)


def test_numpy_idioms_of_user_scenarios_trace_without_forks():
    """What user files write instead of the reference's sqrt(sum(square())): np.linalg.norm, np.dot, np.mean, np.clip, np.maximum /
    minimum, np.min / max, np.where, trigonometry -- traced as nodes (no fork per element) and reproduced exactly."""
    class S(_Base):
        def reward(self, agent, world):
            d = [np.linalg.norm(agent.state.p_pos - l.state.p_pos) for l in world.landmarks]
            spread = np.mean([np.dot(a.state.p_vel, a.state.p_vel) for a in world.agents])
            heading = np.arctan2(agent.state.p_vel[1], agent.state.p_vel[0] + 1e-3)
            wall = np.sum(np.maximum(np.abs(agent.state.p_pos) - 0.9, 0.0))
            return -np.min(d) + 0.1 * np.max(d) - spread - 2.0 * wall + 0.01 * np.cos(heading) + np.sin(agent.state.p_pos[0]) * 0.0 \
                + float(np.where(np.array([True]), 1.0, 2.0)[0]) * 0.0 + np.sum(np.where(agent.state.p_pos > 0.5, agent.state.p_pos, 0.0))

        def observation(self, agent, world):
            rel = np.clip(world.landmarks[0].state.p_pos - agent.state.p_pos, -0.5, 0.5)
            return np.concatenate([agent.state.p_vel, np.minimum(agent.state.p_pos, 0.8), rel, np.tanh(agent.state.p_vel)])
    sc = S()
    t = symtrace.trace(sc)
    # (the only forks: `p_pos > 0.5` -- NumPy's own `>` on an object array asks each element's comparison for its truth)
    assert t.paths["obs"] == [1, 1] and t.paths["rew"] == [4, 4]
    assert symtrace.verify(sc, t, worlds=300) <= 1e-15
    ops = set(n.op for n in symtrace.topo(t.rew + [n for row in t.obs for n in row]))
    assert {"min", "max", "atan2", "cos", "tanh", "ite"} <= ops
    # ... and as device code
    src = symtrace.hip_source(t)
    assert "atan2f(" in src and "cosf(" in src and "tanhf(" in src
    assert np.maximum is not None and np.maximum(1, 2) == 2 and np.clip(5, 0, 1) == 1          # NumPy is itself again


def test_rows_filled_slice_by_slice_rounding_and_number_conversions_trace():
    """`row = np.zeros(n); row[:2] = agent.state.p_pos` (a float array cannot hold symbolic values: np.zeros / ones / empty / full
    give object arrays while a file is traced), np.floor / ceil / rint / round / sign / mod / power / hypot, float(test) /
    int(test) / int(x), a plain number sharing an array with symbolic ones under np.sqrt -- reproduced exactly, no fork."""
    import math

    class S(_Base):
        def reward(self, agent, world):
            d = [np.hypot(*(agent.state.p_pos - l.state.p_pos)) for l in world.landmarks]
            near = min(d)
            score = 0.25 * int(near < 0.3) + float(near > 1.0) - np.round(near, 1) + round(near * 3.0) * 0.01
            mixed = np.sqrt(np.array([near + 1.0, 4.0]))             # (traced: an object array whose 4.0 is a Python float without .sqrt())
            score += mixed[0] * 0.0 + (mixed[1] - 2.0)
            score += int(agent.state.p_pos[0] * 3.0) * 0.1 + math.ceil(agent.state.p_pos[1]) * 0.01 + 2.0 ** agent.state.p_vel[0] * 0.0
            assert isinstance(0.5, float) and isinstance(3, int) and not isinstance(3, float) and float("2.5") == 2.5 and int("7") == 7
            return score - (agent.state.p_pos[0] // 0.5) * 0.01

        def observation(self, agent, world):
            row = np.zeros(12)
            row[0:2] = agent.state.p_pos
            row[2:4] = np.floor(agent.state.p_pos * 4.0) / 4.0
            row[4:6] = np.mod(agent.state.p_pos + 1.0, 0.5)
            row[6:8] = np.sign(agent.state.p_vel + 0.1) * np.power(np.abs(agent.state.p_pos), 1.5)
            row[8:10] = np.rint(agent.state.p_pos * 2.0) + np.ceil(agent.state.p_pos) + np.trunc(agent.state.p_pos * 3.0)
            rest = np.ones(2) * 0.5 + np.full(2, 0.25) + np.zeros_like(agent.state.p_vel) + np.empty(2) * 0.0
            row[10:12] = rest + np.square(agent.state.p_vel) + np.exp(np.zeros(2)) - np.sqrt(np.ones(2))
            return row
    sc = S()
    before = (np.zeros, np.sqrt, np.floor, np.sign, np.full)
    t = symtrace.trace(sc)
    assert (np.zeros, np.sqrt, np.floor, np.sign, np.full) == before and np.zeros(2).dtype == np.float64 and float is not symtrace.sym_float
    assert "float" not in type(sc).reward.__globals__ or type(sc).reward.__globals__["float"] is float
    assert max(t.paths["obs"] + t.paths["rew"]) == 1
    assert symtrace.verify(sc, t, worlds=400) == 0.0
    ops = set(n.op for n in symtrace.topo(t.rew + [n for row in t.obs for n in row]))
    assert {"floor", "rint", "mod", "pow"} <= ops
    src = symtrace.hip_source(t)
    assert "floorf(" in src and "rintf(" in src and "fmodf(" in src and "powf(" in src
    # the staircase functions count as decisions: worlds near a step are found by decision_margin (compared outside the band)
    B = 4000
    P, V, Cw = symtrace.random_states(t, B, np.random.RandomState(3))
    roots = [n for row in t.obs for n in row]
    m = symtrace.decision_margin(roots, B, P=P, V=V, Cw=Cw)
    near = m < 1e-3
    assert 0 < near.sum() < B // 4
    a = np.stack(symtrace.evaluate(roots, B, P=P, V=V, Cw=Cw), axis=1)
    b = np.stack(symtrace.evaluate(roots, B, P=P.astype(np.float32), V=V.astype(np.float32), Cw=Cw, dtype=np.float32), axis=1)
    assert np.abs(a - b)[m > 2e-6].max() < 1e-5


def test_array_constructors_are_the_files_own_and_compiled_library_code_keeps_numpys():
    """The object-array constructors are bound in the FILE's namespace (its `np` is a proxy while it is traced), not patched into
    numpy: numpy.random's compiled code fills what np.empty(n) returns through a C pointer -- with an object array there it wrote
    doubles over object pointers (a segmentation fault when a host reset first called the real np.random.randn under the patches)."""
    seen = {}

    class S(_Base):
        def observation(self, agent, world):
            fixed = np.random.RandomState(7).rand(2)               # compiled code allocating through numpy's own np.empty
            row = np.zeros(6)                                      # the file's: can hold symbolic values
            seen["row"], seen["fixed"], seen["np"] = row.dtype, fixed.dtype, type(np).__name__
            row[0:2] = agent.state.p_pos
            row[2:4] = fixed
            row[4:6] = np.ones_like(fixed) * agent.state.p_vel
            return row
    sc = S()
    t = symtrace.trace(sc)
    assert symtrace.verify(sc, t, worlds=50) == 0.0
    assert seen["row"] == np.float64 and seen["np"] == "module"          # (the last call was verify's concrete one: numpy itself)
    seen.clear()
    with symtrace.injected_builtins(sc):
        row = S.observation.__globals__["np"].zeros(3)
        assert row.dtype == object and type(S.observation.__globals__["np"]).__name__ == "_NumpyProxy"
        import sys
        numpy = sys.modules["numpy"]          # (this test module IS the file here: its own `np` is the proxy right now)
        assert numpy.zeros(3).dtype == numpy.float64 and numpy.random.RandomState(1).randn(4).dtype == numpy.float64      # numpy: untouched
    assert S.observation.__globals__["np"] is np


_FLOAT32_FILE = '''
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.agents = [Agent() for _ in range(2)]
        for i, a in enumerate(world.agents):
            a.name, a.silent = "agent %d" % i, True
        world.landmarks = [Landmark() for _ in range(3)]
        for l in world.landmarks:
            l.movable, l.collide = False, False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for e in world.agents + world.landmarks:
            e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            e.state.p_vel = np.zeros(world.dim_p)
        for a in world.agents:
            a.state.c = np.zeros(world.dim_c)

    def reward(self, agent, world):
        d = np.array([np.linalg.norm(agent.state.p_pos - l.state.p_pos) for l in world.landmarks], dtype=np.float32)
        if np.isnan(d).any() or not np.all(np.isfinite(d)):
            return 0.0
        bonus = 1.0 if np.allclose(agent.state.p_pos, world.landmarks[0].state.p_pos, atol=0.2) else 0.0
        same = 0.5 if np.array_equal(agent.state.p_pos, world.landmarks[1].state.p_pos) else 0.0
        return float(-np.minimum.reduce(d) + np.float32(0.1) * np.float32(d[1]) + bonus + same)

    def observation(self, agent, world):
        parts = [agent.state.p_vel, agent.state.p_pos] + [l.state.p_pos - agent.state.p_pos for l in world.landmarks]
        return np.concatenate(parts).astype(np.float32)          # the gym habit
'''


def test_float32_observations_and_nan_guards(tmp_path):
    """`np.concatenate(parts).astype(np.float32)`, `np.array(..., dtype=np.float32)`, `np.float32(x)`: a node that rounds to single
    precision (exactly what the file computes; a no-op on the device, which is fp32 anyway); np.isnan / isfinite guards,
    np.allclose, np.array_equal, np.minimum.reduce: nodes, no fork."""
    path = tmp_path / "float32.py"
    path.write_text(_FLOAT32_FILE)
    sc = mpe.scenarios.load(str(path)).Scenario()
    t = symtrace.trace(sc)
    assert t.predicated and max(t.paths["obs"] + t.paths["rew"]) == 1
    # the observation (converted once, at the end) bit for bit; the reward, which the file computes IN float32 (rounded after every
    # operation where the graph rounds only at the conversions), to single-precision noise
    assert symtrace.verify(sc, t, worlds=200) < 1e-7
    ops = set(n.op for n in symtrace.topo(t.rew + [n for row in t.obs for n in row]))
    assert "f32" in ops
    src = symtrace.hip_source(t)
    assert "f32" not in src.split("traced_obs", 1)[1]                   # (no code for it)
    B = 2000
    P, V, Cw = symtrace.random_states(t, B, np.random.RandomState(1))
    rows, rew, _ = _host_run_generated(t, P.astype(np.float32).astype(np.float64), V.astype(np.float32).astype(np.float64), Cw, np.zeros((B, 0), np.int64), tmp_path, "f32")
    roots = [n for row in t.obs for n in row] + list(t.rew)
    Pf, Vf = P.astype(np.float32).astype(np.float64), V.astype(np.float32).astype(np.float64)
    vals = symtrace.evaluate(roots, B, P=Pf, V=Vf, Cw=Cw, K=np.zeros((B, 0), np.int64))
    ok = symtrace.decision_margin(roots, B, P=Pf, V=Vf, Cw=Cw, K=np.zeros((B, 0), np.int64)) > 2e-6
    want = np.stack(vals[:len(t.obs[0])], axis=1)
    assert np.abs(rows[0][ok] - want[ok]).max() <= 1e-5 and np.abs(rew[ok, 0] - vals[-2][ok]).max() <= 1e-5


_HELPERS_MODULE = '''import numpy as np

def dist(a, b):
    return np.sqrt(np.sum(np.square(a.state.p_pos - b.state.p_pos)))

def penalty(d):
    if d < 0.2:
        return 1.0
    return 0.0

def row_of(agent, world):
    out = np.zeros(4)
    out[:2] = agent.state.p_pos
    out[2:] = agent.state.p_vel
    return out
'''

_STRUCTURED_FILE = '''import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario
import helpers
from helpers import dist

RADIUS = 0.3


def inside(p, r=RADIUS):
    if abs(p[0]) < r and abs(p[1]) < r:
        return True
    return False


class Base(BaseScenario):
    n_agents = 3

    def make_world(self):
        world = World()
        world.agents = [Agent() for _ in range(self.n_agents)]
        for i, a in enumerate(world.agents):
            a.name, a.silent = "agent %d" % i, True
        world.landmarks = [Landmark() for _ in range(2)]
        for l in world.landmarks:
            l.movable, l.collide = False, False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for e in world.agents + world.landmarks:
            e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            e.state.p_vel = np.zeros(world.dim_p)
        for a in world.agents:
            a.state.c = np.zeros(world.dim_c)

    @staticmethod
    def closeness(a, b):
        return np.exp(-dist(a, b))

    @property
    def weight(self):
        return 0.5

    def reward(self, agent, world):
        r = -sum(dist(agent, l) for l in world.landmarks) * self.weight
        r -= sum(helpers.penalty(dist(agent, o)) for o in world.agents if o is not agent)
        r += 1.0 if inside(agent.state.p_pos) else 0.0
        return r + self.closeness(agent, world.landmarks[0])


class Scenario(Base):
    n_agents = 4

    def observation(self, agent, world):
        return np.concatenate([helpers.row_of(agent, world)] + [l.state.p_pos - agent.state.p_pos for l in world.landmarks])
'''


def test_helper_modules_next_to_the_file_inheritance_staticmethods_and_properties(tmp_path):
    """A scenario split over files: `import helpers` / `from helpers import dist` (the user's own module next to the scenario file),
    a Scenario that inherits from a base class, @staticmethod, @property, module-level functions with `if`s.  The helper module gets
    the same treatment as the file -- the injected names while tracing, a predicated twin -- so `np.zeros` rows and `if d < 0.2`
    inside it neither fail nor fork."""
    import sys
    (tmp_path / "helpers.py").write_text(_HELPERS_MODULE)
    (tmp_path / "structured.py").write_text(_STRUCTURED_FILE)
    before = list(sys.path)
    try:
        sc = mpe.scenarios.load(str(tmp_path / "structured.py")).Scenario()
        t = symtrace.trace(sc)
        assert t.predicated and t.A == 4 and max(t.paths["obs"] + t.paths["rew"]) == 1
        assert symtrace.verify(sc, t, worlds=150) == 0.0
        f = symtrace.trace(sc, predicate=False)                        # (without twins: the helper's `if` forks, once per other agent)
        assert min(f.paths["rew"]) > 8 and symtrace.verify(sc, f, worlds=50) == 0.0
        import helpers
        assert helpers.np is np and "float" not in helpers.__dict__    # the helper module is itself again
    finally:
        sys.path[:] = before
        sys.modules.pop("helpers", None)


def test_what_is_still_not_modelled_falls_back_with_the_reason():
    class Heavy(_Base):
        def observation(self, agent, world):
            return np.heaviside(agent.state.p_pos, 0.5)

    class Close(_Base):
        def reward(self, agent, world):
            return float(np.sum(np.isclose(agent.state.p_pos, world.landmarks[0].state.p_pos, equal_nan=True)))

    class Arc(_Base):
        def reward(self, agent, world):
            return float(np.arcsin(np.clip(agent.state.p_pos[0], -1, 1)))
    for cls, why in ((Heavy, "heaviside"), (Close, "isfinite|isnan"), (Arc, "arcsin of a state-dependent value")):
        with pytest.raises(symtrace.TraceUnsupported, match=why):
            symtrace.trace(cls())


def _host_run_generated(tr, P, V, Cw, K, tmp_path, tag):
    """The generated device functions compiled for the host (tests/c/traced_host.cpp) and run on the given states -> (rows per agent,
    rewards [B, A], dones [B, A])."""
    import subprocess
    src = symtrace.hip_source(tr)
    body = src.split('#include "mpe_internal.h"', 1)[1]                # (the device headers stay out: plain C++ stand-ins in the harness)
    gen = tmp_path / ("gen_%s.h" % tag)
    gen.write_text(body)
    exe = tmp_path / ("traced_host_%s" % tag)
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-DMPE_HOST_TRACED_SOURCE=\"%s\"" % gen,
                        os.path.join(HERE, "c", "traced_host.cpp"), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    B = P.shape[0]
    widths = [len(row) for row in tr.obs]
    with open(tmp_path / ("in_%s.bin" % tag), "wb") as fh:
        fh.write(np.array([B, tr.E, tr.A, tr.dim_c, len(tr.pops), int(getattr(tr, "n_shared", 0))] + widths, np.int32).tobytes())
        fh.write(P.astype(np.float32).tobytes())
        fh.write(V.astype(np.float32).tobytes())
        if tr.dim_c:
            fh.write(Cw.astype(np.float32).tobytes())
        if tr.pops:
            fh.write(K.astype(np.int32).tobytes())
    r = subprocess.run([str(exe), str(tmp_path / ("in_%s.bin" % tag)), str(tmp_path / ("out_%s.bin" % tag))], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    out = np.fromfile(tmp_path / ("out_%s.bin" % tag), np.float32).reshape(B, sum(widths) + 2 * tr.A)
    off = np.cumsum([0] + widths)
    return [out[:, off[i]:off[i + 1]] for i in range(tr.A)], out[:, off[-1]:off[-1] + tr.A], out[:, off[-1] + tr.A:]


@pytest.mark.parametrize("name", ["convoy", "relay", "survey", "herd_info", "simple_tag", "simple_world_comm", "simple_crypto", "nav8", "vector8"])
def test_generated_code_on_the_host_against_the_numpy_evaluation(name, tmp_path):
    """The code GENERATOR without a GPU: traced_obs / traced_shared / traced_rew as symtrace.hip_source writes them, compiled with
    g++ (plain-C++ stand-ins for the device intrinsics) and run on random worlds, against the fp64 NumPy evaluation of the same
    graphs -- fixtures, committed traces of the reference's files, and an 8-agent cooperative-navigation file whose shared reward
    terms go through traced_shared."""
    if name == "herd_info":          # benchmark_data's program (a count over a filtered generator): its rows in the place of the observations
        import copy
        full = symtrace.trace(mpe.scenarios.load(os.path.join(FIXTURES, "herd.py")).Scenario(), want_info=True)
        tr = copy.copy(full)
        tr.obs = [list(row) for row in full.info]
    elif name in ("convoy", "relay", "survey"):
        tr = symtrace.trace(mpe.scenarios.load(os.path.join(FIXTURES, name + ".py")).Scenario())
    elif name == "nav8":
        path = tmp_path / "nav8.py"
        path.write_text(_PREDICATION_FILE.replace("N_AGENTS", "8").replace(
            "        rew = 0\n        for other in world.agents:",
            "        rew = 0\n        for l in world.landmarks:\n            rew -= min(np.linalg.norm(a.state.p_pos - l.state.p_pos) for a in world.agents)\n"
            "        for other in world.agents:"))
        tr = symtrace.trace(mpe.scenarios.load(str(path)).Scenario())
    elif name == "vector8":          # (comparisons of square roots that the shared phase computed)
        path = tmp_path / "vector8.py"
        path.write_text(_VECTOR_FILE.replace("N_AGENTS", "8"))
        tr = symtrace.trace(mpe.scenarios.load(str(path)).Scenario())
    else:
        with open(os.path.join(GOLDEN, "traced_%s.json" % name)) as fh:
            tr = symtrace.from_dict(json.load(fh))
    B = 3000
    rs = np.random.RandomState(8)
    P, V, Cw = symtrace.random_states(tr, B, rs)
    P, V, Cw = P.astype(np.float32).astype(np.float64), V.astype(np.float32).astype(np.float64), Cw.astype(np.float32).astype(np.float64)
    K = np.stack([rs.randint(0, n, B) for n in tr.pops], axis=1) if tr.pops else np.zeros((B, 0), np.int64)
    rows, rew, _ = _host_run_generated(tr, P, V, Cw, K, tmp_path, name)
    if name in ("nav8", "vector8"):
        assert tr.n_shared >= 2                                          # the shared phase is part of what ran
    roots = [n for row in tr.obs for n in row] + list(tr.rew)
    vals = symtrace.evaluate(roots, B, P=P, V=V, Cw=Cw, K=K)
    ok = symtrace.decision_margin(roots, B, P=P, V=V, Cw=Cw, K=K) > 2e-6
    assert ok.mean() > 0.85
    off = np.cumsum([0] + [len(r) for r in tr.obs])
    for i in range(tr.A):
        want = np.stack(vals[off[i]:off[i + 1]], axis=1) if off[i + 1] > off[i] else np.zeros((B, 0))
        assert (np.abs(rows[i][ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))).max(initial=0.0) <= 1e-5, (name, "obs", i)
        w = vals[off[-1] + i]
        assert (np.abs(rew[ok, i] - w[ok]) / np.maximum(1.0, np.abs(w[ok]))).max() <= 1e-5, (name, "rew", i)


_MATH_FILE = '''
import math
from math import exp, hypot
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    def make_world(self):
        world = World()
        world.agents = [Agent() for _ in range(2)]
        for i, a in enumerate(world.agents):
            a.name, a.silent = "agent %d" % i, True
        world.landmarks = [Landmark()]
        world.landmarks[0].movable, world.landmarks[0].collide = False, False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for e in world.agents + world.landmarks:
            e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            e.state.p_vel = np.zeros(world.dim_p)
        for a in world.agents:
            a.state.c = np.zeros(world.dim_c)

    def reward(self, agent, world):
        dx = agent.state.p_pos[0] - world.landmarks[0].state.p_pos[0]
        dy = agent.state.p_pos[1] - world.landmarks[0].state.p_pos[1]
        d = math.sqrt(dx * dx + dy * dy)
        return -hypot(dx, dy) - exp(-d) + 0.1 * math.cos(math.atan2(dy, dx + 1e-2)) + math.tanh(d) * math.fabs(dx) + math.log(1.0 + d) \\
            + math.sqrt(4.0) * 0.0

    def observation(self, agent, world):
        return np.concatenate([agent.state.p_vel, agent.state.p_pos, world.landmarks[0].state.p_pos - agent.state.p_pos])
'''


def test_the_math_module_accepts_symbolic_values_while_a_file_is_traced(tmp_path):
    """`math.sqrt(dx * dx + dy * dy)`, `from math import exp, hypot`: the math module's functions take floats; while a file is traced
    (and while its source is re-executed for predication) they are wrappers that build nodes for symbolic arguments."""
    import math
    path = tmp_path / "mathy.py"
    path.write_text(_MATH_FILE)
    sc = mpe.scenarios.load(str(path)).Scenario()
    before = (math.sqrt, math.exp, math.atan2)
    t = symtrace.trace(sc)
    assert (math.sqrt, math.exp, math.atan2) == before and math.sqrt(9.0) == 3.0
    assert t.predicated and symtrace.verify(sc, t, worlds=200) <= 1e-15
    assert {"sqrt", "exp", "cos", "atan2", "tanh", "log", "abs"} <= set(n.op for n in symtrace.topo(t.rew))


def test_trace_report_tool_says_what_make_env_would_do():
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(HERE), "tools", "trace_report.py")
    r = subprocess.run([sys.executable, tool, os.path.join(FIXTURES, "survey.py")], capture_output=True, text=True)
    assert r.returncode == 0 and "TRACED" in r.stdout and "per-world parameters" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, tool, os.path.join(FIXTURES, "patrol.py")], capture_output=True, text=True)
    assert r.returncode == 1 and "HOST PATH -- scripted agents" in r.stdout, r.stdout + r.stderr


def _pairwise_reward_graph(n_agents):
    """n agents, each reward = an own term minus every pairwise exp(-d2 / 0.3) term (round-5 advisor's reproduction): the
    pairwise terms are shareable sub-expressions under agent-specific accumulation chains."""
    g = symtrace.Graph()
    P = [[g.node("P", (), (i, k)) for k in range(2)] for i in range(n_agents)]
    pair = []
    for i in range(n_agents):
        for j in range(i + 1, n_agents):
            dx, dy = g.binary("sub", P[i][0], P[j][0]), g.binary("sub", P[i][1], P[j][1])
            d2 = g.binary("add", g.binary("mul", dx, dx), g.binary("mul", dy, dy))
            pair.append(g.unary("exp", g.binary("div", g.unary("neg", d2), g.const(0.3))))
    roots = []
    for i in range(n_agents):
        r = g.binary("mul", P[i][0], g.const(float(i + 2)))          # the agent's own term: every chain is agent-specific
        for t in pair:
            r = g.binary("sub", r, t)
        roots.append(r)
    return roots, pair


@pytest.mark.timeout(60)
def test_shared_tasks_terminates_with_more_than_64_shareable_terms():
    """A reward with more than _SHARE_MAX (64) distinct shareable terms under agent-specific chains: 13 agents, 78 pairwise
    terms.  Round 5's loop doubled its cone limit for ever (make_env hung at construction); now the 64 terms that save most
    are shared and the rest stay in every agent's own code.  11 agents / 55 terms: all shared, as before."""
    roots, pair = _pairwise_reward_graph(11)
    tasks = symtrace.shared_tasks(roots, 11)
    assert len(tasks) == 55 and {t.uid for t in tasks} == {t.uid for t in pair}
    roots, pair = _pairwise_reward_graph(13)
    tasks = symtrace.shared_tasks(roots, 13)
    assert len(tasks) == symtrace._SHARE_MAX and {t.uid for t in tasks} <= {t.uid for t in pair}
    assert len({t.uid for t in tasks}) == len(tasks)


def test_graph_constants_keep_the_signed_zero():
    g = symtrace.Graph()
    a, b = g.const(0.0), g.const(-0.0)
    assert a is not b and math.copysign(1.0, a.value) == 1.0 and math.copysign(1.0, b.value) == -1.0
    assert g.const(-0.0) is b and g.const(0.0) is a and g.const(1.5) is g.const(1.5)


def test_keyed_uniform_is_per_world_and_shard_invariant():
    """The torch-path traced reset draws per (seed, global world, episode, draw): distinct triples give distinct streams
    (round-5 advisor: seed s / episode 126 / offset 2209 collided with seed s+1 / 0 / 0) and a shard's worlds draw what the
    same worlds draw inside one big batch."""
    import torch
    ku = refstyle._keyed_uniform
    big = ku(7, 3, 0, 4096, 5, "cpu")
    assert big.shape == (5, 4096) and float(big.min()) >= 0.0 and float(big.max()) < 1.0
    assert torch.equal(ku(7, 3, 1024, 1024, 5, "cpu"), big[:, 1024:2048])
    assert not torch.equal(ku(7, 126, 2209, 64, 5, "cpu"), ku(8, 0, 0, 64, 5, "cpu"))
    assert not torch.equal(ku(7, 4, 0, 64, 5, "cpu"), big[:, :64])
    assert abs(float(big.mean()) - 0.5) < 0.02 and len(torch.unique(big)) > 0.99 * big.numel()
