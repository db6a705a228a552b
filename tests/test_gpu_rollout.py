"""GPU: the two fused-step kernel families agree bit-for-bit, and the fused T-step rollout
(`mpe_rollout_random`) is bit-identical to T x { [mpe_reset]; mpe_random_actions; mpe_step }."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _abi
from multiagent_particle_envs_amd.rollout import RandomRollout, Trajectory
from oracle import spec as ospec
from oracle.mpe_batched import BatchedOracle

pytestmark = pytest.mark.gpu


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("name,kw,B", [("simple", {}, 3000), ("simple_spread", {}, 5000), ("simple_tag", {}, 4097),
                                       ("simple_spread", {"num_agents": 5}, 777),
                                       ("simple_adversary", {}, 2000), ("simple_push", {}, 1111)])
def test_split_and_thread_kernels_bit_identical(name, kw, B):
    rs = np.random.RandomState(1)
    outs = {}
    for impl in ("split", "thread"):
        env = mpe.make_env(name, benchmark=name not in ("simple_adversary", "simple_push"), batch_size=B, **kw)
        env.step_impl = impl
        assert env.fused
        A, E = len(env.world.agents), len(env.world.entities)
        rs = np.random.RandomState(1)
        if env.world.choice_i32 is not None:   # per-world goal landmarks
            env.world.choice_i32.copy_(torch.as_tensor(rs.randint(0, len(env.world.landmarks), size=(1, B)).astype(np.int32)))
        pos = rs.uniform(-1, 1, (B, E, 2)).astype(np.float32)
        pos[::2] *= 0.3
        vel = rs.uniform(-1, 1, (B, A, 2)).astype(np.float32)
        env.world.set_state(pos, vel)
        rec = []
        for t in range(4):
            act = torch.as_tensor(rs.uniform(-1, 1, (A, B, 5)).astype(np.float32)).cuda()
            obs, rew, done, info = env.step(act)
            rec.append([o.clone() for o in obs] + [r.clone() for r in rew] +
                       [x.clone() for tup in info["n"] for x in (tup if isinstance(tup, tuple) else (tup,))
                        if torch.is_tensor(x)])
        rec.append([env.world.pos.clone(), env.world.vel.clone()])
        outs[impl] = rec
    for a, b in zip(outs["split"], outs["thread"]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


@pytest.mark.parametrize("name,kw,B,T,ep", [("simple_spread", {}, 1500, 60, 25), ("simple_tag", {}, 700, 30, 7),
                                            ("simple", {}, 300, 12, 0), ("simple_spread", {"num_agents": 4}, 200, 9, 4),
                                            ("simple_adversary", {}, 900, 30, 5), ("simple_push", {}, 500, 20, 4),
                                            ("simple_spread", {"num_agents": 8}, 130, 11, 4),            # k_multi<ROLL>: several worlds per wave
                                            ("simple_spread", {"num_agents": 20, "num_landmarks": 12}, 70, 7, 3),
                                            ("simple_spread", {"num_agents": 16}, 1000, 8, 3),
                                            ("simple_spread", {"num_agents": 32}, 77, 6, 2),
                                            ("simple_spread", {"num_agents": 7}, 515, 7, 25),             # 8-byte row pieces
                                            ("simple_spread", {"num_agents": 9, "num_landmarks": 22}, 41, 5, 1),
                                            ("simple_spread", {"num_agents": 8}, 70000, 3, 2),            # more groups than the persistent grid
                                            ("simple_spread", {"num_agents": 64}, 37, 6, 5),               # k_duo_roll (two waves per world)
                                            ("simple_spread", {"num_agents": 40}, 130, 9, 3),
                                            ("simple_spread", {"num_agents": 33}, 6, 5, 0),               # odd N: 8-byte row pieces
                                            ("simple_spread", {"num_agents": 64}, 260, 3, 1),             # a reset at every step
                                            ("simple_spread", {"num_agents": 100}, 9, 5, 2),             # > one wave of agents
                                            ("simple_spread", {"num_agents": 3, "num_landmarks": 90}, 21, 5, 3),
                                            # simple_tag at team sizes the reference does not ship: wave-per-world kernel
                                            ("simple_tag", {"num_adversaries": 5, "num_good_agents": 2, "num_landmarks": 1}, 100, 9, 4),
                                            ("simple_tag", {"num_adversaries": 40, "num_good_agents": 30, "num_landmarks": 20}, 10, 5, 2),
                                            # the communication scenarios: words drawn in-kernel, picks re-drawn by in-kernel resets
                                            ("simple_speaker_listener", {}, 700, 13, 4), ("simple_reference", {}, 333, 11, 3),
                                            ("simple_crypto", {}, 200, 9, 4), ("simple_world_comm", {}, 129, 10, 5),
                                            # (team sizes beyond the reference's make_world step through their row programs --
                                            #  env.step / GraphedStep: the fused rollout exists for shapes with a kernel of their own)
                                            # >= 12 MB of rows per step: the rollout kernels built with nontemporal row stores
                                            ("simple_spread", {}, 65536, 3, 2), ("simple_world_comm", {}, 20000, 3, 2)])
def test_fused_rollout_equals_stepwise(name, kw, B, T, ep):
    seed, offset, step0 = 0xABCDEF0123, 4096, 50 if ep in (25, 0) else 14
    # --- stepwise: explicit reset / random_actions / step through the C ABI ----------------------
    env_a = mpe.make_env(name, batch_size=B, seed=seed, **kw)
    env_a.world.world_offset = offset
    env_a._ensure_buffers()
    A = len(env_a.world.agents)
    L = _abi.lib()
    gen = env_a.world.scenario_desc(_abi.MPE_SCN_GENERIC)
    act = torch.zeros((A, B, 5), device="cuda")
    speakers = sum(1 << i for i, a in enumerate(env_a.world.agents) if not a.silent) if env_a._comm is not None else 0
    dim_c = int(env_a.world.dim_c)
    lr = env_a.scenario.landmark_range
    start_pos = env_a.world.pos.clone()
    start_vel = env_a.world.vel.clone()
    start_choice = env_a.world.choice_i32.clone() if env_a.world.choice_i32 is not None else None
    want_obs, want_rew = [], []
    b = env_a._sets[0].bufs
    for t in range(T):
        gt = step0 + t
        if ep and gt % ep == 0:
            _abi.check(L.mpe_reset(C.byref(gen), C.byref(b), B, None, lr, seed, gt // ep, offset, stream()))
        _abi.check(L.mpe_random_actions(act.data_ptr(), None, A, B, seed, gt, offset, stream()))
        if speakers:   # the agents that speak say a uniform random word: rows of comm [A][B][dim_c]
            _abi.check(L.mpe_random_comm(env_a._comm.data_ptr(), A, B, dim_c, speakers, seed, gt, 1, offset, stream()))
        b.act, b.ids, b.u = act.data_ptr(), None, None
        _abi.check(L.mpe_step(C.byref(env_a._desc), C.byref(b), B, stream()))
        want_obs.append([o.clone() for o in env_a._sets[0].obs_n])
        want_rew.append(env_a._sets[0].rew.clone())
    # --- fused: one launch, trajectory outputs -----------------------------------------------------
    env_b = mpe.make_env(name, batch_size=B, seed=seed, **kw)
    env_b.world.world_offset = offset
    env_b.world.pos.copy_(start_pos)
    env_b.world.vel.copy_(start_vel)
    if start_choice is not None:
        env_b.world.choice_i32.copy_(start_choice)
    roll = RandomRollout(env_b, episode_len=ep, pool=2, seed=seed)
    roll.t = step0
    traj = Trajectory(env_b, T)
    roll.fused(T, traj)
    torch.cuda.synchronize()
    for t in range(T):
        for i in range(A):
            assert torch.equal(traj.obs[t][i], want_obs[t][i]), (t, i)
        assert torch.equal(traj.rew[t], want_rew[t]), t
        assert not traj.done[t].any()
    assert torch.equal(env_b.world.pos, env_a.world.pos) and torch.equal(env_b.world.vel, env_a.world.vel)
    if start_choice is not None:   # in-kernel resets re-drew the goals exactly as mpe_reset does
        assert torch.equal(env_b.world.choice_i32, env_a.world.choice_i32)
    if speakers:                   # the agents' comm state after the rollout = their last words
        assert torch.equal(env_b._comm, env_a._comm) and float(env_b._comm.sum()) == B * bin(speakers).count("1")
        sample = RandomRollout(env_a, episode_len=ep, pool=3, seed=seed)   # the pooled stepwise driver draws the same words
        from oracle import philox
        want = philox.one_hot(philox.comm_ids(seed, B, 2, A, dim_c, world_offset=offset), dim_c)
        for i in range(A):
            if (speakers >> i) & 1:
                assert np.array_equal(sample.pool_c[2][i].cpu().numpy(), want[i])
            else:
                assert not sample.pool_c[2][i].any()
    # --- overwrite mode leaves the last step's outputs in the env's buffers --------------------------
    env_c = mpe.make_env(name, batch_size=B, seed=seed, **kw)
    env_c.world.world_offset = offset
    env_c.world.pos.copy_(start_pos)
    env_c.world.vel.copy_(start_vel)
    if start_choice is not None:
        env_c.world.choice_i32.copy_(start_choice)
    roll_c = RandomRollout(env_c, episode_len=ep, pool=2, seed=seed)
    roll_c.t = step0
    out = roll_c.fused(T)
    for i in range(A):
        assert torch.equal(out.obs_n[i], want_obs[-1][i])


def test_graph_replay_matches_eager():
    """The captured-graph rollout (bench --mode graph) produces what the eager enqueue produces."""
    B = 4096
    envs = [mpe.make_env("simple_spread", batch_size=B, seed=5) for _ in range(2)]
    rolls = [RandomRollout(e, episode_len=25, pool=4) for e in envs]
    rolls[0].enqueue(50)
    g = rolls[1].capture(50)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(envs[0].world.pos, envs[1].world.pos)
    for s in range(2):
        assert torch.equal(envs[0]._sets[s].obs, envs[1]._sets[s].obs)


def test_regenerated_moves_match_the_fused_rollout():
    """Fresh moves every step, drawn by `mpe_random_actions` in front of each step on the step's
    stream, eager and as a captured graph: the same trajectory as the fused
    rollout, which draws its moves in-kernel (more steps than the pool has tensors, so cycling would differ)."""
    B, T = 4096, 60
    envs = [mpe.make_env("simple_spread", batch_size=B, seed=5) for _ in range(3)]
    fused = RandomRollout(envs[0], episode_len=25, pool=2)
    fused.fused(T)
    eager = RandomRollout(envs[1], episode_len=25, pool=4, regenerate=True)
    eager.enqueue(T)
    graph = RandomRollout(envs[2], episode_len=25, pool=4, regenerate=True)
    g = graph.capture(T)
    g.replay()
    torch.cuda.synchronize()
    for e in envs[1:]:
        assert torch.equal(e.world.pos, envs[0].world.pos) and torch.equal(e.world.vel, envs[0].world.vel)
        assert torch.equal(e._sets[(T - 1) & 1].obs, envs[0]._sets[0].obs)


def test_rollout_outputs_against_oracle():
    """A free-running fused rollout is still the reference's physics: replay its first steps in the
    fp64 oracle from the trajectory's own observations (positions are in the observation)."""
    B, T = 512, 3
    env = mpe.make_env("simple_spread", batch_size=B, seed=9)
    roll = RandomRollout(env, episode_len=25, pool=2)
    traj = Trajectory(env, T)
    roll.fused(T, traj)
    from oracle import philox
    spec = ospec.simple_spread(3)
    pos0 = philox.reset_positions(9, B, 0, 3, 3, 1.0)
    orc = BatchedOracle(spec, B)
    orc.set_state(pos0, np.zeros((B, 3, 2)))
    for t in range(T):
        act = philox.one_hot(philox.action_ids(9, B, t, 3))
        obs64, rew64, _, _ = orc.step(act)
        for i in range(3):
            got = traj.obs[t][i].cpu().numpy()
            assert np.abs(got - obs64[i]).max() < 2e-5          # free-running, 3 steps
        assert np.abs(traj.rew[t].cpu().numpy() - rew64).max() < 1e-4


def test_fresh_moves_rollout_started_mid_block_reads_the_right_moves():
    """RandomRollout(regenerate=True): the pool is one block of `pool` consecutive global steps; a rollout whose step
    counter is moved (here: into the middle of block 3) must redraw that block, not read stale tensors -- every step's
    moves are exactly mpe_random_actions(step)."""
    B, P = 512, 5
    env = mpe.make_env("simple_spread", batch_size=B, seed=8)
    roll = RandomRollout(env, episode_len=0, pool=P, regenerate=True)
    roll.t = 17
    ref = mpe.make_env("simple_spread", batch_size=B, seed=8)
    ref.world.pos.copy_(env.world.pos)
    ref.world.vel.copy_(env.world.vel)
    act = torch.zeros((3, B, 5), device="cuda")
    for t in range(17, 24):          # crosses from block 3 into block 4
        _abi.check(_abi.lib().mpe_random_actions(act.data_ptr(), None, 3, B, 8, t, 0, stream()))
        ref.step(act)
    roll.enqueue(7)
    torch.cuda.synchronize()
    assert torch.equal(env.world.pos, ref.world.pos) and torch.equal(env.world.vel, ref.world.vel)


def test_rollout_with_integer_action_ids_equals_env_steps_with_those_ids():
    """RandomRollout(action_ids=True): the moves reach mpe_step as int32 ids [A][B] (`discrete_action_input`,
    environment.py:161-167) -- eager and as a captured graph the same trajectory as env.step() fed the ids
    mpe_random_actions writes, which are the NumPy Philox restatement's; and NOT the one-hot trajectory (Q3: id 1 is -x)."""
    from oracle import philox
    B, T, P, EP = 1000, 11, 4, 7
    envs = [mpe.make_env("simple_tag", batch_size=B, seed=21) for _ in range(4)]
    ref = envs[0]
    ref.discrete_action_input = True
    ref._ensure_buffers()
    gen = ref.world.scenario_desc(_abi.MPE_SCN_GENERIC)
    ids = torch.zeros((4, B), dtype=torch.int32, device="cuda")
    for t in range(T):
        if t % EP == 0:
            _abi.check(_abi.lib().mpe_reset(C.byref(gen), C.byref(ref._sets[0].bufs), B, None, ref.scenario.landmark_range,
                                            21, t // EP, 0, stream()))
        _abi.check(_abi.lib().mpe_random_actions(None, ids.data_ptr(), 4, B, 21, t, 0, stream()))
        assert np.array_equal(ids.cpu().numpy(), philox.action_ids(21, B, t, 4))
        ref_obs, ref_rew, _, _ = ref.step([ids[i] for i in range(4)])
    eager = RandomRollout(envs[1], episode_len=EP, pool=P, regenerate=True, action_ids=True)
    out_e = eager.enqueue(T)
    graph = RandomRollout(envs[2], episode_len=EP, pool=P, regenerate=True, action_ids=True)
    graph.capture(T).replay()
    onehot = RandomRollout(envs[3], episode_len=EP, pool=P, regenerate=True)
    onehot.enqueue(T)
    torch.cuda.synchronize()
    for e in envs[1:3]:
        assert torch.equal(e.world.pos, ref.world.pos) and torch.equal(e.world.vel, ref.world.vel)
    for i in range(4):
        assert torch.equal(out_e.obs_n[i], ref_obs[i]) and torch.equal(out_e.rew[i], ref_rew[i])
    assert not torch.equal(envs[3].world.pos, ref.world.pos)


@pytest.mark.parametrize("compiled", [False, True])
def test_rollouts_whose_episodes_end_by_the_programs_done_tests(compiled):
    """examples/corral.py with an arena (a done_spec) + a horizon + auto_reset: ONE mpe_rollout_rows_episode launch for T steps --
    per world, the episode ends where an agent leaves the arena or the horizon is reached, the world restarts inside that step --
    against the same T steps as mpe_step_rows_episode launches (RandomRollout.enqueue) and as plain env.step calls with the same
    block-drawn moves: every step's rows, rewards and dones, the counters, the picks, the state -- to the bit."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_rowspec as tr
    B, T = 1500, 14

    def make():
        e = tr.corral_env(B, arena=0.95, max_episode_steps=6, auto_reset=True)
        if compiled:
            assert e.compile_program()
        e.reset()
        return e
    d, e, f = make(), make(), make()
    assert d._episode_in_launch and d._prog.has_done and d.program_compiled == compiled
    rd, re_ = RandomRollout(d, episode_len=0, pool=4, regenerate=True), RandomRollout(e, episode_len=0, pool=4, regenerate=True)
    traj = Trajectory(d, T)
    rd.fused(T, traj)
    L = _abi.lib()
    moves = torch.empty((4, f.n, B, 5), device="cuda")
    ended = 0
    for t in range(T):
        out_t = re_.enqueue(1)
        if t % 4 == 0:
            _abi.check(L.mpe_random_actions_block(moves.data_ptr(), None, f.n, B, rd.seed, t, 4, 0, _abi.raw_stream(f.world.device)), "draw")
        of, rf, df, _ = f.step(moves[t % 4].clone())
        for i in range(d.n):
            assert torch.equal(traj.obs[t][i], out_t.obs_n[i]) and torch.equal(traj.obs[t][i], of[i]), (t, i)
            assert torch.equal(traj.rew[t][i], out_t.reward_n[i]) and torch.equal(traj.rew[t][i], rf[i]), (t, i)
            assert torch.equal(traj.done[t][i], out_t.done_n[i]) and torch.equal(traj.done[t][i], df[i]), (t, i)
        ended += int(traj.done[t].any(dim=0).sum())
    for x in (e, f):
        assert torch.equal(d.world.pos, x.world.pos) and torch.equal(d.world.vel, x.world.vel)
        assert torch.equal(d.episode_step, x.episode_step) and torch.equal(d.world.choice_i32, x.world.choice_i32)
        assert d.world._episode == x.world._episode
    assert ended > B            # every world ended at least once (the horizon), many earlier
    with pytest.raises(_abi.MpeError, match="episode_len = 0"):
        RandomRollout(make(), episode_len=5)


@pytest.mark.parametrize("B", [1, 5, 70, 129])
def test_program_rollouts_at_batch_sizes_that_do_not_fill_a_wave(B):
    """Fewer worlds than a wave, and a ragged last workgroup: the fused rollout of a row-program env (interpreted and compiled in)
    == its per-step launches."""
    envs = [mpe.make_env("simple_adversary", batch_size=B, num_agents=4, num_adversaries=2, seed=2, compile_program=False) for _ in range(3)]
    assert envs[2].compile_program()
    rolls = [RandomRollout(e, episode_len=3, pool=3, regenerate=True) for e in envs]
    T = 7
    trajs = [Trajectory(e, T) for e in envs[1:]]
    rolls[1].fused(T, trajs[0])
    rolls[2].fused(T, trajs[1])
    for t in range(T):
        out = rolls[0].enqueue(1)
        for tr_ in trajs:
            for i in range(envs[0].n):
                assert torch.equal(tr_.obs[t][i], out.obs_n[i]) and torch.equal(tr_.rew[t][i], out.reward_n[i]), (B, t, i)
    for e in envs[1:]:
        assert torch.equal(e.world.pos, envs[0].world.pos) and torch.equal(e.world.vel, envs[0].world.vel)


@pytest.mark.parametrize("compiled", [False, True])
def test_fused_rollout_of_a_program_env_whose_leader_speaks(compiled):
    """simple_world_comm at a team size without a kernel of its own (its leader says a drawn word every step): one
    mpe_rollout_rows launch == the per-step launches (mpe_random_comm + mpe_step_rows), rows, rewards, state, the comm state."""
    B, T = 640, 11
    def make():
        e = mpe.make_env("simple_world_comm", batch_size=B, num_good_agents=2, num_adversaries=3, seed=4, compile_program=False)
        if compiled:
            assert e.compile_program()
        return e
    d, e = make(), make()
    assert d._prog is not None and d._has_speakers
    rd, re_ = RandomRollout(d, episode_len=4, pool=4, regenerate=True), RandomRollout(e, episode_len=4, pool=4, regenerate=True)
    assert rd.speakers == 1
    traj = Trajectory(d, T)
    rd.fused(T, traj)
    for t in range(T):
        out_t = re_.enqueue(1)
        for i in range(d.n):
            assert torch.equal(traj.obs[t][i], out_t.obs_n[i]), (t, i, float((traj.obs[t][i] - out_t.obs_n[i]).abs().max()))
            assert torch.equal(traj.rew[t][i], out_t.reward_n[i]), (t, i)
    assert torch.equal(d.world.pos, e.world.pos) and torch.equal(d.world.vel, e.world.vel) and torch.equal(d._comm, e._comm)
    assert float(d._comm[0].sum()) == B and float(d._comm[1:].abs().sum()) == 0.0      # one word per world from the leader, nobody else


@pytest.mark.parametrize("compiled", [False, True])
def test_a_row_program_env_rolls_out_through_per_step_launches(compiled):
    """RandomRollout on a row-program env (a team size without a kernel of its own): enqueue() == the same resets, moves and
    env.step calls by hand; a captured graph replays the same steps; the fused T-step launch stays with the built-ins."""
    B, T = 768, 12
    def make():
        e = mpe.make_env("simple_adversary", batch_size=B, num_agents=4, num_adversaries=2, seed=9, compile_program=False)
        if compiled:
            assert e.compile_program()
        return e
    a, b, c = make(), make(), make()
    assert a.fused and a._prog is not None and a.program_compiled == compiled
    ra, rc = RandomRollout(a, episode_len=5, pool=5, regenerate=True), RandomRollout(c, episode_len=5, pool=5, regenerate=True)
    out = ra.enqueue(T)
    # by hand: reset every 5 steps with the rollout's episode numbers, the same block-drawn moves, env.step
    L = _abi.lib()
    moves = torch.empty((5, b.n, B, 5), device="cuda")
    b._ensure_buffers()
    for t in range(T):
        if t % 5 == 0:
            _abi.check(L.mpe_random_actions_block(moves.data_ptr(), None, b.n, B, ra.seed, t, 5, 0, _abi.raw_stream(b.world.device)), "draw")
            bufs = b._sets[0].bufs
            _abi.check(L.mpe_reset(C.byref(ra._gen_desc), C.byref(bufs), B, None, ra._lr, ra.seed, t // 5, 0, _abi.raw_stream(b.world.device)), "reset")
        ob, rb, _, _ = b.step(moves[t % 5].clone())
    assert torch.equal(a.world.pos, b.world.pos) and torch.equal(a.world.vel, b.world.vel)
    for i in range(a.n):
        assert torch.equal(out.obs_n[i], ob[i]) and torch.equal(out.reward_n[i], rb[i]), i
    g = rc.capture(T)
    rc.t = 0
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(c.world.pos, a.world.pos) and torch.equal(c.world.vel, a.world.vel)
    # ... and the fused T-step launch (mpe_rollout_rows: state in LDS, moves and resets drawn in the kernel): every step's rows,
    # rewards and dones in its own trajectory block, the state after the last step -- the per-step launches' to the bit
    d, e = make(), make()
    rd, re_ = RandomRollout(d, episode_len=5, pool=5, regenerate=True), RandomRollout(e, episode_len=5, pool=5, regenerate=True)
    traj = Trajectory(d, T)
    rd.fused(T, traj)
    for t in range(T):
        out_t = re_.enqueue(1)
        for i in range(d.n):
            assert torch.equal(traj.obs[t][i], out_t.obs_n[i]), (t, i, float((traj.obs[t][i] - out_t.obs_n[i]).abs().max()))
            assert torch.equal(traj.rew[t][i], out_t.reward_n[i]), (t, i)
            assert not traj.done[t][i].any()
    assert torch.equal(d.world.pos, e.world.pos) and torch.equal(d.world.vel, e.world.vel)
    assert torch.equal(d.world.choice_i32, e.world.choice_i32)
    rd.fused(3)                       # without a trajectory: over the env's output set 0; the clock goes on (steps 12, 13, 14)
    last = re_.enqueue(3)
    assert torch.equal(d.world.pos, e.world.pos) and all(torch.equal(d._sets[0].obs_n[i], last.obs_n[i]) for i in range(d.n))
    with pytest.raises(_abi.MpeError, match="episode clock"):
        RandomRollout(mpe.make_env("simple_adversary", batch_size=64, num_agents=4, num_adversaries=2, max_episode_steps=5), episode_len=5)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [("simple_spread", {}), ("simple_adversary", {}), ("simple_reference", {})])
def test_reset_and_move_block_in_one_launch_equals_the_two_launches(name, kw):
    """An episode that starts where a block of moves starts: `mpe_reset_random_actions_block` (reset_world + the block's moves, one
    launch) against `mpe_reset` + `mpe_random_actions_block` -- the same draws, so the same worlds, moves and outputs, bit for bit."""
    B, T = 3000, 60
    envs = [mpe.make_env(name, batch_size=B, seed=4, **kw) for _ in range(2)]
    rolls = [RandomRollout(e, episode_len=25, pool=25, regenerate=True) for e in envs]
    rolls[1].fuse_reset_and_draw = False
    for r in rolls:
        r.enqueue(T)
    torch.cuda.synchronize()
    a, b = envs
    assert torch.equal(a.world.pos, b.world.pos) and torch.equal(a.world.vel, b.world.vel)
    assert torch.equal(rolls[0].pool_t, rolls[1].pool_t)
    if a.world.choice_i32 is not None:
        assert torch.equal(a.world.choice_i32, b.world.choice_i32)
    for s_ in range(2):
        assert torch.equal(a._sets[s_].obs, b._sets[s_].obs) and torch.equal(a._sets[s_].rew, b._sets[s_].rew)
