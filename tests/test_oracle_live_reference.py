"""Build container only: the oracle against the reference RUN LIVE, on seeds and actions the committed fixtures do not hold.

tests/golden/*.npz pin the oracle to recordings; this test closes the remaining gap ("the oracle fits the fixtures") by
recording fresh worlds from the unmodified /root/reference in a subprocess (tests/golden/gen_golden.py's `record`, the same
recorder that made the fixtures, with the shape-only gym stub) -- seeds derived from the run's date, so every day tests
new worlds -- and replaying them through the fp64 batched oracle and the per-object loop port.  /root/reference does not
exist on the GPU box: there the test is skipped (nothing on the GPU side reads the reference).
"""
import datetime
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import spec as ospec
from oracle.mpe_batched import BatchedOracle, seeded_initial_state
from oracle.mpe_loop import LoopEnv

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "multiagent", "core.py")),
                                reason="the reference tree exists in the build container only")

RECORDER = r"""
import sys, numpy as np
sys.path.insert(0, %(golden)r)
import gen_golden as gg
base = int(sys.argv[1])
jobs = {
    "simple": gg.record("simple", gg.make_env("simple"), [base + k for k in range(6)], 30),
    "simple_spread": gg.record("simple_spread", gg.make_env("simple_spread", benchmark=True), [base + 10 + k for k in range(9)], 20,
                               squeeze_every=3, arng=np.random.RandomState(base)),
    "simple_tag": gg.record("simple_tag", gg.make_env("simple_tag", benchmark=True), [base + 30 + k for k in range(9)], 20,
                            squeeze_every=3, squeeze=0.25, arng=np.random.RandomState(base + 1)),
    "simple_spread_n7": gg.record("simple_spread_n7", gg.spread_n(7), [base + 50 + k for k in range(4)], 8, squeeze_every=2,
                                  squeeze=0.4, arng=np.random.RandomState(base + 2)),
}
for name, data in jobs.items():
    np.savez(sys.argv[2] + "/" + name + ".npz", **data)
"""
TOL = 1e-12


def _base_seed():
    """The day's seed (yymmdd), or MPE_LIVE_SEED=<yymmdd> to replay the worlds of the day a failure was seen on."""
    return int(os.environ.get("MPE_LIVE_SEED") or datetime.date.today().strftime("%y%m%d"))


def _close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    scale = np.maximum(1.0, np.abs(b))
    assert np.all(np.abs(a - b) <= TOL * scale), "max scaled err %.3e (replay these worlds with MPE_LIVE_SEED=%d)" % (
        float(np.max(np.abs(a - b) / scale)), _base_seed())


@pytest.fixture(scope="module")
def fresh(tmp_path_factory):
    out = tmp_path_factory.mktemp("live_reference")
    base = _base_seed() * 100     # new worlds every day, reproducible within one (and replayable: MPE_LIVE_SEED)
    env = dict(os.environ, SUPPRESS_MA_PROMPT="1", PYTHONDONTWRITEBYTECODE="1", PYTHONWARNINGS="ignore")
    r = subprocess.run([sys.executable, "-c", RECORDER % {"golden": os.path.join(ROOT, "tests", "golden")}, str(base), str(out)],
                       capture_output=True, text=True, env=env, timeout=600, cwd=str(out))
    assert r.returncode == 0, r.stderr[-3000:]
    return lambda name: dict(np.load(os.path.join(str(out), name + ".npz")))


CASES = [("simple", ospec.simple(), False), ("simple_spread", ospec.simple_spread(3), True), ("simple_tag", ospec.simple_tag(), True),
         ("simple_spread_n7", ospec.simple_spread(7), True)]


@pytest.mark.parametrize("name,spec,bench", CASES, ids=[c[0] for c in CASES])
def test_batched_oracle_replays_fresh_reference_worlds(name, spec, bench, fresh):
    g = fresh(name)
    T, W, A = g["rew"].shape
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


@pytest.mark.parametrize("name,spec,bench", CASES, ids=[c[0] for c in CASES])
def test_c_restatement_replays_fresh_reference_worlds(name, spec, bench, fresh):
    """oracle/mpe_oracle.c (bench.py's `c_port`), teacher-forced from the reference's own states."""
    from oracle import build_c
    g = fresh(name)
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


def test_seeded_resets_of_fresh_worlds(fresh):
    """reset_world consumes the global MT19937 stream in the reference's order: the un-squeezed worlds' initial
    states follow from their seeds alone."""
    for name, spec, every in (("simple", ospec.simple(), 0), ("simple_spread", ospec.simple_spread(3), 3), ("simple_tag", ospec.simple_tag(), 3)):
        g = fresh(name)
        plain = [w for w in range(len(g["seeds"])) if not every or w % every != every - 1]
        pos, vel = seeded_initial_state(spec, g["seeds"][plain])
        assert np.array_equal(pos, g["pos0"][plain]) and np.array_equal(vel, g["vel0"][plain])


def test_loop_port_replays_fresh_reference_worlds(fresh):
    """oracle/mpe_loop.py (bench.py's CPU baseline, kind "port") on the same fresh worlds, one world at a time."""
    for name, spec in (("simple_spread", ospec.simple_spread(3)), ("simple_tag", ospec.simple_tag())):
        g = fresh(name)
        T, W, A = g["rew"].shape
        for w in range(0, W, 2):
            env = LoopEnv(spec)
            env.set_state(g["pos0"][w], g["vel0"][w])
            for t in range(T):
                obs, rew, done, _ = env.step([g["act"][t, w, i] for i in range(A)])
                for i in range(A):
                    _close(obs[i], g["obs%d" % i][t, w])
                _close(np.array(rew, dtype=np.float64), g["rew"][t, w])


# ---- the six other scenarios: the product's own reset_world / observation / reward (torch, on the CPU) against the
#      reference run live -- the check tests/test_f3_scenarios.py makes against the committed f3_*.npz, on fresh seeds ----------
RECORDER_F3 = r"""
import sys, numpy as np
sys.path.insert(0, %(golden)r)
import gen_golden_scenarios as gs
base = int(sys.argv[1])
for k, (name, sq) in enumerate([("simple_adversary", 0), ("simple_push", 0), ("simple_speaker_listener", 0), ("simple_reference", 0),
                                ("simple_crypto", 0), ("simple_world_comm", 8)]):
    data = gs.record(name, [base + 100 * k + j for j in range(64)], 6, squeeze_every=sq, stage=True)
    np.savez(sys.argv[2] + "/f3_" + name + ".npz", **data)
import gen_golden_shapes as gsh      # other team sizes: the reference's callbacks on resized agent lists
for k, (name, A, nadv) in enumerate(gsh.SHAPES):
    data = gs.record(name, [base + 1000 + 100 * k + j for j in range(32)], 4, squeeze_every=8 if name == "simple_world_comm" else 0,
                     stage=True, env_factory=gsh.factory(name, A, nadv))
    np.savez(sys.argv[2] + "/shape_" + name + "_" + str(A) + "_" + str(nadv) + ".npz", **data)
"""


@pytest.fixture(scope="module")
def fresh_f3(tmp_path_factory):
    out = tmp_path_factory.mktemp("live_reference_f3")
    base = _base_seed() * 1000 + 7
    env = dict(os.environ, SUPPRESS_MA_PROMPT="1", PYTHONDONTWRITEBYTECODE="1", PYTHONWARNINGS="ignore")
    r = subprocess.run([sys.executable, "-c", RECORDER_F3 % {"golden": os.path.join(ROOT, "tests", "golden")}, str(base), str(out)],
                       capture_output=True, text=True, env=env, timeout=900, cwd=str(out))
    assert r.returncode == 0, r.stderr[-3000:]
    return lambda name: dict(np.load(os.path.join(str(out), name + ".npz")))


def test_host_callbacks_of_the_other_six_scenarios_on_fresh_reference_worlds(fresh_f3):
    import test_f3_scenarios as f3
    for name in f3.NAMES:
        f3.test_reset_and_callbacks_match_reference_on_cpu(name, fresh_f3)


def test_f3_oracle_replays_fresh_reference_worlds(fresh_f3):
    """oracle/mpe_f3.py (the fp64 restatement the GPU tests compare the six other scenarios' fused kernels with) against
    the reference run live: 64 fresh worlds per scenario, staged like the fixtures, whole trajectories at 1e-12, and
    the seeded resets (np.random.choice picks, then agents, then landmarks)."""
    import test_oracle_golden as tg
    from oracle.mpe_f3 import seeded_initial_state_f3
    for name in tg.F3:
        g = fresh_f3("f3_" + name)
        spec = ospec.by_name(name)
        tg._replay_f3(spec, g)
        plain = np.flatnonzero(~g["staged"])
        pos, vel, choice = seeded_initial_state_f3(spec, g["seeds"])
        assert np.array_equal(choice, g["choice"]), "MPE_LIVE_SEED=%d" % _base_seed()
        assert np.array_equal(pos[plain], g["pos0"][plain]), "MPE_LIVE_SEED=%d" % _base_seed()


def test_f3_oracle_at_other_team_sizes_replays_fresh_reference_worlds(fresh_f3):
    """The team-size variants of simple_adversary / simple_world_comm (tests/golden/gen_golden_shapes.py) on fresh seeds."""
    import test_oracle_golden as tg
    from oracle.mpe_f3 import seeded_initial_state_f3
    for name, A, nadv in tg.SHAPES:
        g = fresh_f3("shape_%s_%d_%d" % (name, A, nadv))
        spec = tg.shape_spec(name, A, nadv)
        tg._replay_f3(spec, g)
        plain = np.flatnonzero(~g["staged"])
        pos, vel, choice = seeded_initial_state_f3(spec, g["seeds"])
        assert np.array_equal(choice, g["choice"]), "MPE_LIVE_SEED=%d" % _base_seed()
        assert np.array_equal(pos[plain], g["pos0"][plain]), "MPE_LIVE_SEED=%d" % _base_seed()
