"""The step server (include/mpe_hip.h: mpe_step_server_*; rollout.StepServer): per-step COMMANDS to one resident launch.

The served steps must be the per-step kernel's, bit for bit -- T x {mpe_reset at the episode boundaries; mpe_step} through
RandomRollout.enqueue on a second env with the same seed and the same move tensors -- in every commanding pattern: all doorbells
rung ahead (the pipelined regime the server exists for), one at a time with the host in between (the closed loop: the server is
AHEAD of its commander and must publish a step's completion without waiting for the next command), in bursts; on ragged batches,
with per-world picks (simple_adversary) and an immovable-free / adversary mix (simple_tag).  A server whose doorbell never rings
gives up after its timeout and says so; shapes it cannot serve are refused with the reason."""
import time

import numpy as np
import pytest
import torch

import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _abi
from multiagent_particle_envs_amd.rollout import RandomRollout, StepServer

pytestmark = pytest.mark.gpu


def reference_steps(name, kw, B, T, EP, ring):
    """T steps through the per-step launches: -> (the move ring, per step: obs rows of every agent, rew, done, pos, vel)."""
    env = mpe.make_env(name, batch_size=B, seed=3, **kw)
    rr = RandomRollout(env, episode_len=EP, pool=ring, regenerate=False)
    out = []
    for _ in range(T):
        o = rr.enqueue(1)
        torch.cuda.synchronize()
        pos, vel = env.world.pos.clone(), env.world.vel.clone()
        ch = env.world.choice_i32.clone() if env.world.choice_i32 is not None else None
        out.append(([x.clone() for x in o.obs_n], o.rew.clone(), o.done.clone(), pos, vel, ch))
    comm = rr.pool_c.clone() if rr.pool_c is not None else None
    return (rr.pool_t.clone(), comm, env._comm.clone() if env._comm is not None else None), out


def same(step, srv, env, g, what):
    obs, rew, done, pos, vel, ch = step
    o_s, r_s, d_s = srv.outputs(g)
    for i, (a, b) in enumerate(zip(o_s, obs)):
        assert torch.equal(a, b), "%s: obs of agent %d at step %d" % (what, i, g)
    assert torch.equal(r_s, rew) and torch.equal(d_s, done), "%s: rew / done at step %d" % (what, g)


# (up to 1.5 workgroups per CU -- 24 576 worlds on 256 CUs -- the server is the DUAL-role kernel, a physics and a rows wave per
#  agent; beyond, the single-role one: the 32 768- and 40 000-world cases)
CASES = [("simple_spread", {}, 4096, 25), ("simple_spread", {}, 1000, 7), ("simple_tag", {}, 1000, 25),
         ("simple_adversary", {}, 777, 10), ("simple_push", {}, 640, 25), ("simple", {}, 130, 5),
         ("simple_spread", {"num_agents": 5}, 900, 25), ("simple_spread", {}, 32768, 25), ("simple_tag", {}, 40000, 9),
         # the communication scenarios: the speakers' words come from an utterance ring as the moves from the move ring
         ("simple_speaker_listener", {}, 1000, 25), ("simple_reference", {}, 700, 6), ("simple_crypto", {}, 1500, 25),
         ("simple_world_comm", {}, 600, 10)]


@pytest.mark.parametrize("name,kw,B,EP", CASES)
def test_served_steps_are_the_launched_steps_bit_for_bit(name, kw, B, EP):
    T, ring = (60, 16) if B < 20000 else (28, 8)
    (moves, comm, comm_after), ref = reference_steps(name, kw, B, T, EP, ring)
    env = mpe.make_env(name, batch_size=B, seed=3, **kw)
    srv = StepServer(env, moves, slots=T, episode_len=EP, timeout_s=5.0, comm=comm)
    srv.start(T)
    srv.ring(T)                      # every step commanded ahead: the server runs them back to back
    srv.join()
    torch.cuda.synchronize()
    srv.check()
    assert int(srv.flag.min()) == T and int(srv.door.item()) == T
    for g in range(T):
        same(ref[g], srv, env, g, "all rung ahead")
    assert torch.equal(env.world.pos, ref[-1][3]) and torch.equal(env.world.vel, ref[-1][4])      # the state after the last step, in HBM
    if ref[-1][5] is not None:
        assert torch.equal(env.world.choice_i32, ref[-1][5])
    if comm_after is not None:       # the agents' comm state afterwards = their last words
        assert torch.equal(env._comm, comm_after)


@pytest.mark.parametrize("name,kw,B,EP", [CASES[1], CASES[3], CASES[7], CASES[11]])
def test_closed_loop_commands_one_step_at_a_time(name, kw, B, EP):
    """ring -> wait -> read -> ring ...: the server is ahead of its commander at every step (the idle path: a step's completion is
    published without the next command), state visible in HBM after each step, host pauses in between."""
    T, ring = 12, 4
    (moves, comm, _), ref = reference_steps(name, kw, B, T, EP, ring)
    env = mpe.make_env(name, batch_size=B, seed=3, **kw)
    srv = StepServer(env, moves, slots=2, episode_len=EP, timeout_s=20.0, comm=comm)
    srv.start(T)
    for g in range(T):
        srv.ring()
        srv.wait()
        torch.cuda.current_stream().synchronize()         # (only this stream: the server launch is still running on its own)
        assert int(srv.status.item()) == 0
        same(ref[g], srv, env, g, "closed loop")
        assert torch.equal(env.world.pos, ref[g][3]) and torch.equal(env.world.vel, ref[g][4]), "state in HBM after step %d" % g
        if g in (2, 7):
            time.sleep(0.05)
    srv.join()
    torch.cuda.synchronize()
    srv.check()


def test_bursts_and_two_launches():
    """Doorbells in bursts with the host asleep in between, over two consecutive server launches (the second starts from the
    state the first left in HBM; doorbell and flags count on)."""
    name, kw, B, EP = "simple_spread", {}, 2048, 25
    T, ring = 50, 8
    (moves, _, _), ref = reference_steps(name, kw, B, T, EP, ring)
    env = mpe.make_env(name, batch_size=B, seed=3, **kw)
    srv = StepServer(env, moves, slots=T, episode_len=EP, timeout_s=20.0)
    srv.start(30)
    for n in (1, 3, 9, 17):
        srv.ring(n)
        srv.wait()
        torch.cuda.current_stream().synchronize()
        time.sleep(0.01)
    srv.start(20)
    srv.ring(20)
    srv.join()
    torch.cuda.synchronize()
    srv.check()
    for g in range(T):
        same(ref[g], srv, env, g, "bursts")
    assert torch.equal(env.world.pos, ref[-1][3])


def test_a_server_nobody_commands_gives_up():
    env = mpe.make_env("simple_spread", batch_size=512, seed=3)
    moves = torch.zeros((2, 3, 512, _abi.MPE_ACTION_DIM), device="cuda")
    srv = StepServer(env, moves, slots=2, timeout_s=0.2)
    srv.start(3)
    srv.ring(2)
    t0 = time.time()
    srv.join()
    torch.cuda.synchronize()
    assert time.time() - t0 < 5.0
    assert int(srv.status.item()) == 1 and int(srv.flag.min()) == 2      # two steps served, then the timeout
    with pytest.raises(_abi.MpeError, match="timed out"):
        srv.check()
    with pytest.raises(_abi.MpeError, match="start"):
        srv.ring(5)


def test_what_the_server_refuses():
    env = mpe.make_env("simple_speaker_listener", batch_size=256)
    with pytest.raises(_abi.MpeError, match="utterances"):      # a scenario whose agents speak needs the utterance ring
        StepServer(env, torch.zeros((2, 2, 256, _abi.MPE_ACTION_DIM), device="cuda"))
    env = mpe.make_env("simple_spread", batch_size=256, num_agents=10)      # 20 entities: no wave-per-agent kernel
    with pytest.raises(_abi.MpeError, match="step server"):
        StepServer(env, torch.zeros((2, 10, 256, _abi.MPE_ACTION_DIM), device="cuda"))
    env = mpe.make_env("simple_spread", batch_size=256)
    with pytest.raises(_abi.MpeError, match="moves"):
        StepServer(env, torch.zeros((2, 3, 255, _abi.MPE_ACTION_DIM), device="cuda"))
    B = 1 << 20                                        # 16 384 workgroups cannot all be resident
    env = mpe.make_env("simple_spread", batch_size=B)
    srv = StepServer(env, torch.zeros((1, 3, B, _abi.MPE_ACTION_DIM), device="cuda"), slots=1)
    with pytest.raises(_abi.MpeError, match="resident"):
        srv.start(1)


@pytest.mark.parametrize("name", ["simple_spread", "simple_reference"])
@pytest.mark.parametrize("graphs", [False, True])
def test_served_rollout_is_the_fresh_moves_rollout(graphs, name):
    """rollout.ServedRollout (bench.py's step-server leg): block draws into the halves of a 2-episode move ring, 25 doorbells per
    episode, one server launch per enqueue -- the state and the last step's outputs equal RandomRollout(regenerate=True)'s
    launches.  With graphs=True the caller-side half of an episode is a HIP graph per ring half (its draw repeats the moves of
    episodes 0 / 1): compared over the first two episodes, then run on for the protocol's sake."""
    from multiagent_particle_envs_amd.rollout import ServedRollout
    B, EP = 4096, 25
    K = 2 * EP if graphs else 4 * EP
    ref_env = mpe.make_env(name, batch_size=B, seed=5)
    rr = RandomRollout(ref_env, episode_len=EP, pool=EP, regenerate=True)
    o = rr.enqueue(K)
    torch.cuda.synchronize()
    env = mpe.make_env(name, batch_size=B, seed=5)
    roll = ServedRollout(env, episode_len=EP, slots=2, graphs=graphs)
    roll.enqueue(K)
    torch.cuda.synchronize()
    roll.srv.check()
    assert torch.equal(env.world.pos, ref_env.world.pos) and torch.equal(env.world.vel, ref_env.world.vel)
    obs, rew, done = roll.srv.outputs(K - 1)
    for a, b in zip(obs, o.obs_n):
        assert torch.equal(a, b)
    assert torch.equal(rew, o.rew) and int(roll.srv.flag.min()) == K and int(roll.srv.door.item()) == K
    if graphs:
        roll.enqueue(6 * EP)
        torch.cuda.synchronize()
        roll.srv.check()
        assert int(roll.srv.flag.min()) == K + 6 * EP and bool(torch.isfinite(env.world.pos).all())


@pytest.mark.parametrize("name,kw,B", [("simple_spread", {}, 3000), ("simple_tag", {}, 4096), ("simple_spread", {}, 40000)])
def test_step_many_is_the_env_step_loop(name, kw, B):
    """rollout.step_many(env, moves[T]): the caller's `for t: env.step(moves[t])` loop as ONE launch on the caller's own stream (the
    doorbell rung T ahead, then the server launch: no second stream) -- bit-identical to the T env.step calls, twice in a row."""
    from multiagent_particle_envs_amd.rollout import step_many
    T = 9
    ref = mpe.make_env(name, batch_size=B, seed=11, **kw)
    env = mpe.make_env(name, batch_size=B, seed=11, **kw)
    ref.reset()
    env.world.set_state(*ref.world.get_state())
    A = ref.n
    g = torch.Generator(device="cpu").manual_seed(5)
    moves = torch.empty((T, A, B, _abi.MPE_ACTION_DIM), device="cuda")
    for rnd in range(2):
        moves.copy_(torch.nn.functional.one_hot(torch.randint(0, 5, (T, A, B), generator=g), 5).float())
        outs = step_many(env, moves)
        torch.cuda.synchronize()
        for t in range(T):
            obs_n, rew_n, done_n, _ = ref.step(moves[t])
            o_s, r_s, d_s = outs[t]
            for i in range(A):
                assert torch.equal(o_s[i], obs_n[i]), (rnd, t, i)
                assert torch.equal(r_s[i], rew_n[i]) and torch.equal(d_s[i], done_n[i])
        assert torch.equal(env.world.pos, ref.world.pos) and torch.equal(env.world.vel, ref.world.vel)


def test_launches_whose_commands_precede_them_need_no_residency():
    """`ahead`: every step of a launch is commanded before the launch starts (ring, then start, in stream order) -- nothing in it
    ever waits, so its 16 384 workgroups need not be resident: step_many and ServedRollout(launch_per_episode=True) at 1 048 576
    worlds, bit-identical to the launched steps."""
    from multiagent_particle_envs_amd.rollout import ServedRollout, step_many
    B, EP = 1 << 20, 5
    ref_env = mpe.make_env("simple_spread", batch_size=B, seed=5)
    rr = RandomRollout(ref_env, episode_len=EP, pool=EP, regenerate=True)
    o = rr.enqueue(3 * EP)
    torch.cuda.synchronize()
    env = mpe.make_env("simple_spread", batch_size=B, seed=5)
    roll = ServedRollout(env, episode_len=EP, slots=2, graphs=False, launch_per_episode=True)
    roll.enqueue(3 * EP)
    torch.cuda.synchronize()
    roll.srv.check()
    assert torch.equal(env.world.pos, ref_env.world.pos) and torch.equal(env.world.vel, ref_env.world.vel)
    obs, rew, done = roll.srv.outputs(3 * EP - 1)
    assert all(torch.equal(a, b) for a, b in zip(obs, o.obs_n)) and torch.equal(rew, o.rew)
    del roll, rr
    moves = torch.nn.functional.one_hot(torch.randint(0, 5, (3, 3, B)), 5).float().cuda()
    outs = step_many(env, moves)
    for t in range(3):
        obs_n, rew_n, _, _ = ref_env.step(moves[t])
        assert all(torch.equal(a, b) for a, b in zip(outs[t][0], obs_n)) and torch.equal(outs[t][1][0], rew_n[0])


@pytest.mark.parametrize("what", ["spread10", "spread40", "tag12", "corral", "convoy"])
def test_step_many_beyond_the_server(what):
    """env.step_many where the step server does not serve: simple_spread / simple_tag beyond 16 entities (the wave-per-world
    rollouts with the caller's moves: mpe_rollout_actions) and row-program envs -- a user scenario (examples/corral.py) and a
    traced reference-style file (tests/refstyle/convoy.py): mpe_rollout_rows_actions -- bit-identical to the env.step loop."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name, kw, B = {"spread10": ("simple_spread", {"num_agents": 10}, 3000), "spread40": ("simple_spread", {"num_agents": 40}, 700),
                   "tag12": ("simple_tag", {"num_good_agents": 4, "num_adversaries": 8, "num_landmarks": 3}, 1500),
                   "corral": (os.path.join(root, "examples", "corral.py"), {}, 4096),
                   "convoy": (os.path.join(root, "tests", "refstyle", "convoy.py"), {}, 2000)}[what]
    T = 7
    ref = mpe.make_env(name, batch_size=B, seed=11, **kw)
    env = mpe.make_env(name, batch_size=B, seed=11, **kw)
    ref.reset()
    env.reset()
    assert torch.equal(env.world.pos, ref.world.pos)
    A = ref.n
    g = torch.Generator(device="cpu").manual_seed(6)
    moves = torch.empty((T, A, B, _abi.MPE_ACTION_DIM), device="cuda")
    for rnd in range(2):
        moves.copy_(torch.nn.functional.one_hot(torch.randint(0, 5, (T, A, B), generator=g), 5).float())
        outs = env.step_many(moves)
        torch.cuda.synchronize()
        for t in range(T):
            obs_n, rew_n, done_n, _ = ref.step(moves[t])
            o_s, r_s, d_s = outs[t]
            for i in range(A):
                assert torch.equal(o_s[i], obs_n[i]), (rnd, t, i)
                assert torch.equal(r_s[i], rew_n[i]) and torch.equal(d_s[i], done_n[i])
        assert torch.equal(env.world.pos, ref.world.pos) and torch.equal(env.world.vel, ref.world.vel)
