"""Random-action rollout driver: the throughput workload of BASELINE.json ("steps/sec on
random-action rollouts"), with the host taken out of the per-step loop.

A rollout of K env steps is: every `episode_len` steps a device-side reset (`mpe_reset`,
MADDPG's 25-step episodes: SURVEY.md 8d), and every step one `mpe_step` launch that reads a
one-hot action tensor resident in HBM (a pool of pre-generated uniform random moves,
`mpe_random_actions`) and writes that step's obs / reward / done into the env's output buffers.
Three ways to issue it:
  eager    K C-ABI calls from Python (host-bound below ~5 us per step)
  graph    the same K launches captured once into a HIP graph and replayed (no host in the loop)
  fused    `mpe_rollout_random`: one persistent launch, state kept in registers between steps,
           moves drawn in-kernel (bit-identical to the pool's), outputs still written every step
"""
import ctypes as C
import os

import torch

from . import _abi


class RandomRollout(object):
    def __init__(self, env, episode_len=25, pool=16, seed=None, regenerate=False, action_ids=False):
        """action_ids: the moves are handed to mpe_step as int32 ids [A][B] (the reference's `discrete_action_input`
        format, environment.py:161-167: 4 bytes per agent instead of a 20-byte one-hot row; mind its opposite sign
        convention) instead of one-hot rows -- only for the eager / graph drivers.
        regenerate: False -- the pool's tensors are drawn once and cycled (the policy's output is already resident
        in HBM when the step is launched); True -- every step consumes moves nobody has used before: whenever the pool
        is exhausted (every `pool` steps; use pool = episode_len) ONE `mpe_random_actions_block` launch redraws all of
        it for the next `pool` global steps, on the same stream.  (Per-step redraws cost a second 2.8 us launch per
        step; drawing one step ahead on a side stream costs a fork/join per step inside a HIP graph, which was
        measured slower still: 13.0 vs 9.4 us per step; drawing the next BLOCK on a side stream, one fork/join per
        episode (commit d24e237), is slower too: 8.2-8.3 vs 7.3 us per step at B = 65536, 92.9 vs 85.2 at B = 1M --
        concurrent kernels do not overlap usefully on this stack.  The block draw is one 98 MB-write launch per 25 steps.)"""
        if not env.fused:
            raise _abi.MpeError("RandomRollout drives the device-side step (fused scenarios and row-program envs); this env steps "
                                "through Python callbacks: use env.step / GraphedStep")
        self._prog = getattr(env, "_prog", None)      # a row-program env: mpe_step_rows / mpe_rollout_rows
        # ... whose episodes end on the device (a done_spec and / or max_episode_steps, auto_reset): the rollout leaves the episode ends
        # to it -- every step is mpe_step_rows_episode, T of them one mpe_rollout_rows_episode launch; one episode number per step
        self._in_launch = self._prog is not None and bool(env._episode_in_launch)
        if self._prog is not None and env.max_episode_steps and not self._in_launch:
            raise _abi.MpeError("RandomRollout keeps its own episode clock (episode_len): build the row-program env without "
                                "max_episode_steps, or with auto_reset (the env's done programs / horizon then end the episodes)")
        if self._in_launch and int(episode_len):
            raise _abi.MpeError("this env ends its episodes itself (done programs / max_episode_steps with auto_reset): episode_len = 0")
        if int(episode_len) and not env._device_restart_ok:
            raise _abi.MpeError("RandomRollout's episode resets are drawn on the device as world.reset_uniform(landmark_range, "
                                "choices=choice_pops); this env's reset_world is not known to be that (a built-in scenario, or a "
                                "scenario with `device_reset = True`, in rng_mode 'device'): use episode_len = 0 and reset it yourself")
        if env._py_obs or env._py_reward or env._py_done or env._py_info:
            raise _abi.MpeError("RandomRollout runs on the device and evaluates the built-in callbacks only: this env has "
                                "Python observation / reward / done / info callbacks (use env.step, or GraphedStep)")
        self.env = env
        self.world = env.world
        self.fuse_reset_and_draw = not os.environ.get("MPE_NO_FUSED_RESET_DRAW")      # (False: the two launches, the A/B)
        self.episode_len = int(episode_len)
        self.seed = int(env.world.seed if seed is None else seed) & (2 ** 64 - 1)
        env._ensure_buffers()
        A, B = len(self.world.agents), self.world.batch_size
        self.A, self.B = A, B
        self.action_ids = bool(action_ids)
        if self.action_ids:
            self.pool_t = torch.empty((int(pool), A, B), dtype=torch.int32, device=self.world.device)
        else:
            self.pool_t = torch.empty((int(pool), A, B, _abi.MPE_ACTION_DIM), dtype=torch.float32, device=self.world.device)
        self.pool = [self.pool_t[p] for p in range(int(pool))]
        # communication scenarios: the agents that speak say a uniform random word per step (`mpe_random_comm`), pooled
        # like the moves; comm tensor p holds the words of the global steps t with t % len(pool) == p
        self.speakers = 0
        self.pool_c = None
        if env._comm is not None:
            for i, agent in enumerate(self.world.agents):
                if not agent.silent:
                    self.speakers |= 1 << i
            self.pool_c = torch.zeros((int(pool), A, B, int(self.world.dim_c)), dtype=torch.float32, device=self.world.device)
        self.t = 0          # global step counter (also indexes the Philox action stream)
        self.regenerate = bool(regenerate)
        self._L = _abi.lib()
        self._lr = float(getattr(env._scenario, "landmark_range", 1.0))
        self._gen_desc = self.world.scenario_desc(_abi.MPE_SCN_GENERIC)
        # scenario_desc caches one descriptor: take private copies of both
        self._gen_desc = _copy_struct(self._gen_desc)
        self._desc = _copy_struct(env._desc)
        self._fill_pool()

    def _stream(self):
        return _abi.raw_stream(self.world.device)

    def _fill_pool(self, t0=0, stream=None):
        """Action tensor p holds the moves of global steps t with t % len(pool) == p -- those of steps
        t0 .. t0 + len(pool) - 1 after this call (t0 a multiple of len(pool)); without `regenerate` the pool is
        drawn once and cycled (the fused kernel draws fresh moves every step)."""
        _abi.check(self._L.mpe_random_actions_block(None if self.action_ids else self.pool_t.data_ptr(),
                                                    self.pool_t.data_ptr() if self.action_ids else None,
                                                    self.A, self.B, self.seed, int(t0),
                                                    len(self.pool), int(self.world.world_offset),
                                                    stream if stream is not None else self._stream()),
                   "mpe_random_actions_block")
        if self.pool_c is not None:
            _abi.check(self._L.mpe_random_comm(self.pool_c.data_ptr(), self.A, self.B, int(self.world.dim_c),
                                               self.speakers, self.seed, int(t0), len(self.pool), int(self.world.world_offset),
                                               stream if stream is not None else self._stream()), "mpe_random_comm")
        self._pool_block = int(t0) // len(self.pool)     # which block of global steps the pool holds

    def enqueue(self, steps):
        """Enqueue `steps` env steps (and the resets that fall among them) on the current stream."""
        env, w = self.env, self.world
        L, desc, B = self._L, self._desc, self.B
        st = self._stream()
        for _ in range(steps):
            refill = self.regenerate and self.t // len(self.pool) != self._pool_block   # entering a block the pool does not hold
            boundary = bool(self.episode_len) and self.t % self.episode_len == 0
            if refill and boundary and self.fuse_reset_and_draw and self.t % len(self.pool) == 0 and \
                    (self._prog is None or not self._prog.struct.reset_boxes):
                # an episode starts where a block of moves starts: reset_world and the block's moves in ONE launch (the same draws
                # as mpe_reset + mpe_random_actions_block; one dependent launch less in front of every episode)
                _abi.check(L.mpe_reset_random_actions_block(
                    C.byref(self._gen_desc), C.byref(env._sets[0].bufs), B, self._lr, self.t // self.episode_len,
                    None if self.action_ids else self.pool_t.data_ptr(), self.pool_t.data_ptr() if self.action_ids else None,
                    self.seed, int(self.t), len(self.pool), int(w.world_offset), st), "mpe_reset_random_actions_block")
                if self.pool_c is not None:
                    _abi.check(L.mpe_random_comm(self.pool_c.data_ptr(), self.A, B, int(w.dim_c), self.speakers, self.seed, int(self.t),
                                                 len(self.pool), int(w.world_offset), st), "mpe_random_comm")
                self._pool_block = self.t // len(self.pool)
                refill = boundary = False
            if refill:
                self._fill_pool(self.t // len(self.pool) * len(self.pool), st)
            if boundary:
                b = env._sets[0].bufs
                if self._prog is not None:      # the program's own placement (MpeRowProgram.reset_boxes), as its in-launch restarts draw it
                    _abi.check(L.mpe_reset_rows(C.byref(self._gen_desc), C.byref(b), self._prog.ref, B, None, self._lr, self.seed,
                                                self.t // self.episode_len, int(w.world_offset), st), "mpe_reset_rows")
                else:
                    _abi.check(L.mpe_reset(C.byref(self._gen_desc), C.byref(b), B, None, self._lr, self.seed,
                                           self.t // self.episode_len, int(w.world_offset), st), "mpe_reset")
            out = env._sets[self.t & 1]
            b = out.bufs
            move = self.pool[self.t % len(self.pool)].data_ptr()
            b.act, b.ids = (None, move) if self.action_ids else (move, None)
            b.u = None
            out.act_ptr = None      # MultiAgentEnv.step's fast path must not trust its note of what b.act holds
            if self.pool_c is not None:
                b.comm = self.pool_c[self.t % len(self.pool)].data_ptr()
            if self._in_launch:
                if env.episode_step is None:
                    env.episode_step = torch.zeros(B, dtype=torch.int32, device=w.device)
                _abi.check(L.mpe_step_rows_episode(C.byref(desc), C.byref(b), self._prog.ref, B, env.episode_step.data_ptr(),
                                                   env.max_episode_steps, self._lr, self.seed, int(w._episode), int(w.world_offset), st),
                           "mpe_step_rows_episode")
                w._episode += 1
            elif self._prog is not None:
                _abi.check(L.mpe_step_rows(C.byref(desc), C.byref(b), self._prog.ref, B, st), "mpe_step_rows")
            else:
                _abi.check(L.mpe_step(C.byref(desc), C.byref(b), B, st), "mpe_step")
            self.t += 1
        if self.pool_c is not None and steps > 0:   # the agents' comm state after the last step = their last words
            env._comm.copy_(self.pool_c[(self.t - 1) % len(self.pool)])
            for out in env._sets:
                out.bufs.comm = env._comm.data_ptr()
        if not torch.cuda.is_current_stream_capturing():
            self._mark_stale()     # host-side bookkeeping + an episode_step fill: not part of a captured graph (capture() / replay do it)
        return env._sets[(self.t - 1) & 1]

    def _mark_stale(self):
        """Device-side resets bypass Scenario.reset_world: the env re-derives the scenario's Python-side per-world
        state (MultiAgentEnv.sync_from_device) before its next step() / reset() through the Python API; the env's
        per-world step counters follow the rollout's episode clock."""
        env = self.env
        env._scenario_state_stale = True
        env._fast_acts.clear()      # step()'s fast path re-validates the caller's action tensor after a device-side rollout
        for out in env._sets or ():
            out.act_ptr = None
        if env.episode_step is not None and self.episode_len:
            env.episode_step.fill_(self.t % self.episode_len)
        if self._in_launch:
            # the per-world step counters moved on the device; the host's note of the steps at which a world can reach the
            # horizon (env._may_finish) is no longer true: env.step takes a fresh episode number at every step from here on
            env._horizon_clock_lost = True

    def capture(self, steps):
        """Capture `steps` env steps into a HIP graph (torch.cuda.CUDAGraph); replay() re-runs them.
        `steps` should be a multiple of lcm(episode_len, pool, 2) for the replay to be periodic.  The two warm-up steps
        in front of the capture really run (the world moves on by two steps; the step counter does not): begin the
        captured steps at a reset (self.t a multiple of episode_len) when the trajectory has to be a known one."""
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=self.world.device)
        s.wait_stream(torch.cuda.current_stream(self.world.device))
        t0 = self.t
        with torch.cuda.stream(s):
            self.enqueue(2)          # warm the code objects outside capture
            self.t = t0
            torch.cuda.synchronize()
            if self.regenerate:
                self._pool_block = None      # every block draw of the captured steps is part of the graph
            with torch.cuda.graph(g, stream=s):
                self.enqueue(steps)
        torch.cuda.current_stream(self.world.device).wait_stream(s)
        self._mark_stale()
        return _MarkedGraph(g, self)

    def fused(self, steps, trajectory=None):
        """One `mpe_rollout_random` launch covering `steps` env steps.  With `trajectory` (a
        Trajectory of at least `steps` blocks) every step's outputs land in their own block;
        without it each step overwrites the env's output set 0."""
        if trajectory is None:
            b = self.env._sets[0].bufs
            ret = self.env._sets[0]
        else:
            assert trajectory.T >= steps
            b = trajectory.bufs
            ret = trajectory
        b.act = b.ids = b.u = None
        if trajectory is None:
            ret.act_ptr = None      # (as in enqueue: the env's set 0 no longer points at the caller's action tensor)
        if self._in_launch:             # ... and the episode ends by its done programs / horizon, inside the launch
            w = self.world
            if self.env.episode_step is None:
                self.env.episode_step = torch.zeros(self.B, dtype=torch.int32, device=w.device)
            _abi.check(self._L.mpe_rollout_rows_episode(C.byref(self._desc), C.byref(b), self._prog.ref, self.B, int(steps),
                                                        self.env.episode_step.data_ptr(), self.env.max_episode_steps, self._lr, self.seed,
                                                        self.t, int(w._episode), int(w.world_offset), 1 if trajectory is not None else 0,
                                                        int(self.speakers), self._stream()), "mpe_rollout_rows_episode")
            w._episode += int(steps)
        elif self._prog is not None:    # a row-program env: the same launch shape, the rows and rewards by its program
            _abi.check(self._L.mpe_rollout_rows(C.byref(self._desc), C.byref(b), self._prog.ref, self.B, int(steps), self.episode_len,
                                                self._lr, self.seed, self.t, int(self.world.world_offset),
                                                1 if trajectory is not None else 0, int(self.speakers), self._stream()), "mpe_rollout_rows")
        else:
            _abi.check(self._L.mpe_rollout_random(C.byref(self._desc), C.byref(b), self.B, int(steps), self.episode_len,
                                                  self._lr, self.seed, self.t, int(self.world.world_offset),
                                                  1 if trajectory is not None else 0, self._stream()),
                       "mpe_rollout_random")
        self.t += steps
        self._mark_stale()
        return ret


class _MarkedGraph(object):
    """A captured rollout graph whose replay() also does the host-side bookkeeping of the steps it re-runs (the env's
    scenario state is stale, its per-world step counters follow the rollout's episode clock) -- outside the graph."""

    def __init__(self, graph, *rollouts):
        self.graph, self.rollouts = graph, rollouts

    def replay(self):
        self.graph.replay()
        for r in self.rollouts:
            r._mark_stale()


class StreamedRollout(object):
    """S independent sub-batches of worlds, each advanced on its own HIP stream.

    Worlds never interact, so a batch can be cut into S slices that progress independently: step
    t+1 of slice 0 overlaps step t of slice 1.  A dependent chain of small launches is bound by
    per-launch latency (dispatch, wave ramp-up, the load -> compute -> store chain, drain), not by
    bandwidth; running S chains side by side hides that latency exactly like more waves per SIMD
    hide instruction latency.  Each slice is an ordinary env (own tensors, `world_offset` = its
    first global world), so results are identical to one big batch (shard invariance).
    """

    def __init__(self, rollouts):
        self.rollouts = list(rollouts)
        dev = self.rollouts[0].world.device
        self.device = dev
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.rollouts]

    @property
    def B(self):
        return sum(r.B for r in self.rollouts)

    def _fan(self, fn):
        if len(self.rollouts) == 1:      # nothing to fan out: stay on the caller's stream
            fn(self.rollouts[0])
            return
        cur = torch.cuda.current_stream(self.device)
        for r, s in zip(self.rollouts, self.streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                fn(r)
        for s in self.streams:
            cur.wait_stream(s)

    def enqueue(self, steps):
        self._fan(lambda r: r.enqueue(steps))

    def fused(self, steps, trajectories=None):
        it = iter(trajectories) if trajectories is not None else None
        self._fan(lambda r: r.fused(steps, next(it) if it is not None else None))

    def capture(self, steps):
        g = torch.cuda.CUDAGraph()
        main = torch.cuda.Stream(device=self.device)
        main.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(main):
            t0 = [r.t for r in self.rollouts]
            self.enqueue(2)                      # load code objects outside capture
            for r, t in zip(self.rollouts, t0):
                r.t = t
                if r.regenerate:
                    r._pool_block = None     # every block draw of the captured steps is part of the graph
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=main):
                self.enqueue(steps)
        torch.cuda.current_stream(self.device).wait_stream(main)
        for r in self.rollouts:
            r._mark_stale()
        return _MarkedGraph(g, *self.rollouts)

    def set_episode_len(self, n):
        for r in self.rollouts:
            r.episode_len = n


class Trajectory(object):
    """Device buffers for T consecutive steps' outputs (what a rollout collector keeps for training):
    obs[t][i] is the [B, D_i] observation of agent i after step t, rew[t] / done[t] are [A, B]."""

    def __init__(self, env, T):
        env._ensure_buffers()
        w = env.world
        A, B, dev = len(w.agents), w.batch_size, w.device
        off = env._obs_off
        self.T, self.A, self.B = int(T), A, B
        per = int(off[-1]) * B
        self.obs_flat = torch.zeros(self.T * per, dtype=torch.float32, device=dev)
        self.obs = [[self.obs_flat[t * per + off[i] * B: t * per + off[i + 1] * B].view(B, off[i + 1] - off[i])
                     for i in range(A)] for t in range(self.T)]
        self.rew = torch.zeros((self.T, A, B), dtype=torch.float32, device=dev)
        self.done = torch.zeros((self.T, A, B), dtype=torch.bool, device=dev)
        b = _abi.MpeBuffers()
        b.pos, b.vel = w.pos.data_ptr(), w.vel.data_ptr()
        b.obs, b.rew, b.done = self.obs_flat.data_ptr(), self.rew.data_ptr(), self.done.data_ptr()
        if env._entity_table is not None:
            b.entity_table = env._entity_table.data_ptr()
        if w.choice_i32 is not None:
            b.choice = w.choice_i32.data_ptr()
        if env._comm is not None:
            b.comm = env._comm.data_ptr()
        self.bufs = b


def _copy_struct(s):
    c = type(s)()
    C.memmove(C.byref(c), C.byref(s), C.sizeof(s))
    return c


class StepServer(object):
    """Per-step COMMANDS to one resident launch instead of a launch per step (include/mpe_hip.h: the step server).

        srv = StepServer(env, moves)          # moves: [ring, A, B, 5] one-hot tensors; step g reads moves[g % ring]
        srv.start(T)                          # one launch on the server's own stream: up to T steps, gated by the doorbell
        for g in range(T):
            ... write moves[g % ring] on the current stream ...
            srv.ring()                        # "step g may run": a one-thread launch on the CURRENT stream
        srv.wait()                            # the current stream continues when every commanded step's outputs are in memory
        out = srv.outputs(g)                  # views of output block g % slots: obs_n (list of [B, D_i]), rew [A, B], done [A, B]

    The steps are the per-step kernel's, bit for bit (tests/test_gpu_server.py); episodes restart inside the launch every
    `episode_len` global steps (mpe_reset's draws).  The server wins where the moves of step g + 1 exist before step g has
    finished -- its steps then run back to back at the kernel's span, without the 1-2 us dependent-launch gap; a closed loop pays
    a ring and a wait launch per step and is better served by env.step.  One server per env at a time; the env's own outputs
    (env.step's ping-pong sets) are not touched: a served step's outputs live in the server's blocks."""

    def __init__(self, env, moves, slots=2, episode_len=0, seed=None, timeout_s=2.0, probe_graph=False, probe=True, ahead=False,
                 comm=None):
        """comm: the communication scenarios' utterances, [ring, A, B, dim_c] one-hot rows (the speaking agents' rows are read;
        step g reads comm[g % ring], as moves[g % ring])."""
        if not env.fused or getattr(env, "_prog", None) is not None:
            raise _abi.MpeError("StepServer serves the fused built-in scenarios (the shapes with a wave-per-agent kernel)")
        if env._py_obs or env._py_reward or env._py_done or env._py_info:
            raise _abi.MpeError("StepServer evaluates the built-in callbacks only")
        self.env, self.world = env, env.world
        env._ensure_buffers()
        w = self.world
        A, B = len(w.agents), w.batch_size
        self._L = _abi.lib()
        self._desc = _copy_struct(env._desc)
        if self._L.mpe_step_server_supported(C.byref(self._desc), B) != 1:
            raise _abi.MpeError("no step server for this scenario / shape (the shapes with a wave-per-agent kernel: <= 16 entities)")
        if int(episode_len) and not env._device_restart_ok:
            raise _abi.MpeError("StepServer's in-launch resets are world.reset_uniform's device draws: this env's reset_world is not that")
        moves = moves if moves.dim() == 4 else moves[None]
        if tuple(moves.shape[1:]) != (A, B, _abi.MPE_ACTION_DIM) or moves.dtype != torch.float32 or not moves.is_contiguous() \
                or moves.device != w.device:
            raise _abi.MpeError("moves: a contiguous float32 [ring, A, B, %d] tensor on the env's device" % _abi.MPE_ACTION_DIM)
        self.moves, self.A, self.B = moves, A, B
        self.comm = None
        if env._comm is not None:      # a communication scenario: the utterance ring next to the move ring
            if comm is None or tuple(comm.shape) != (int(moves.shape[0]), A, B, int(w.dim_c)) or comm.dtype != torch.float32 \
                    or not comm.is_contiguous() or comm.device != w.device:
                raise _abi.MpeError("comm: a contiguous float32 [ring, A, B, %d] tensor of utterances on the env's device "
                                    "(this scenario's agents speak)" % int(w.dim_c))
            self.comm = comm
        self.slots, self.episode_len = int(slots), int(episode_len)
        self.seed = int(w.seed if seed is None else seed) & (2 ** 64 - 1)
        self._lr = float(getattr(env._scenario, "landmark_range", 1.0))
        self.blocks = Trajectory(env, self.slots)
        dev = w.device
        self.door = torch.zeros(1, dtype=torch.int64, device=dev)
        self.flag = torch.zeros(int(self._L.mpe_step_server_flags(B)), dtype=torch.int64, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        s = _abi.MpeStepServer()
        s.door, s.flag, s.status = self.door.data_ptr(), self.flag.data_ptr(), self.status.data_ptr()
        s.act_ring, s.ring, s.slots, s.timeout_us = moves.data_ptr(), int(moves.shape[0]), self.slots, int(timeout_s * 1e6)
        s.ahead = 1 if ahead else 0      # every launch's commands precede it (run(), per-episode launches): any batch size
        s.comm_ring = self.comm.data_ptr() if self.comm is not None else None
        self._srv = s
        self._commander = _abi.raw_stream(dev).value      # commands come from the stream that is current NOW
        if probe:
            self.stream = self._concurrent_stream(dev, graph=probe_graph)
        else:       # (every command will precede its launch -- tools/server_profile.py under a profiler that serialises dispatches)
            self.stream, self.stream_probe = torch.cuda.Stream(device=dev), None
        self.launch_events = None
        self.t = 0              # global step at which the next start() begins (absolute: doorbell and flags count from 0)
        self.commanded = 0
        self.served_to = 0      # steps covered by the launches started so far
        torch.cuda.synchronize(dev)      # (the words above are zero before anything can ring)

    def _concurrent_stream(self, dev, probe_timeout_s=0.05, n=40, graph=False):
        """A stream whose launches run CONCURRENTLY with the commanding (current) stream's, at full rate: HIP multiplexes its
        streams onto a few hardware queues.  A doorbell queued behind the resident server on the SAME hardware queue never starts;
        with some pairings the doorbells ARE processed, but one per ~33 us instead of one per 1.7 us (seen with doorbells replayed
        from a HIP graph: a graph's kernels keep a tie to the stream they were captured on, so capture, replay and probe all use
        the commanding stream).  Candidates are probed with the
        library's own wait / ring pair in the server's pattern: a wait launch on the candidate spins on a scratch word until n
        one-thread rings on the commanding stream -- eager, or with graph=True replayed from a graph captured on it, as
        ServedRollout issues them -- have counted it up; the rings' span, from events on the commanding stream, is the verdict."""
        word = torch.zeros(1, dtype=torch.int64, device=dev)
        st = torch.zeros(1, dtype=torch.int32, device=dev)
        p = _abi.MpeStepServer()
        p.door = p.flag = word.data_ptr()
        p.status, p.timeout_us = st.data_ptr(), int(probe_timeout_s * 1e6)
        raw = lambda: _abi.raw_stream(dev)      # noqa: E731

        def rings():
            for _ in range(n):
                _abi.check(self._L.mpe_step_server_ring(C.byref(p), 1, raw()), "mpe_step_server_ring (probe)")
        g = None
        if graph:
            rings()                             # (code object loaded outside the capture)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=torch.cuda.current_stream(dev)):
                rings()
        tried = []
        for k in range(12):
            # (normal priority only: with the server on a HIGH-priority stream 3 of 10 instances served one dispatch of the
            #  commanding stream per ~33 us -- 32 us per step with a doorbell per step, 5.8 instead of 2.0 with one per episode --
            #  although this probe passed; 0 of 20 with normal-priority streams.  profiles/r6_step_server.txt; MPE_SERVER_PRIORITY=1
            #  brings the high-priority candidates back for the A/B.)
            hp = (k % 2 == 0) and os.environ.get("MPE_SERVER_PRIORITY", "0") == "1"
            cand = torch.cuda.Stream(device=dev, priority=-1) if hp else torch.cuda.Stream(device=dev)
            span = []
            for attempt in range(2):            # (the first round also loads the code objects)
                word.zero_()
                st.zero_()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(cand):
                    _abi.check(self._L.mpe_step_server_wait(C.byref(p), 1, n, raw()), "mpe_step_server_wait (probe)")
                e0.record()
                if g is not None:
                    g.replay()
                else:
                    rings()
                e1.record()
                torch.cuda.synchronize(dev)
                span.append(e0.elapsed_time(e1) * 1e3 / n)
            ok = int(st.item()) == 0 and span[-1] < 8.0        # us per ring: 1.7 from a graph, 3-5 eager (host-bound), ~33 when sliced
            tried.append({"candidate": k, "us_per_ring": round(span[-1], 2), "timed_out": int(st.item()) != 0})
            if ok:
                self.stream_probe = {"picked": k, "graph_rings": bool(graph), "candidates": tried}
                return cand
        raise _abi.MpeError("StepServer: no stream runs concurrently with the commanding one at full rate (12 candidates probed: %r): "
                            "the doorbells could not keep up with the resident server" % (tried,))

    def run(self, T):
        """T steps whose moves ALREADY EXIST (moves[g % ring] of the next T global steps): the doorbell first, then the launch, both
        on the CURRENT stream -- every command precedes the launch, so nothing has to overtake anything: no second stream, no
        probe, nothing resident that waits.  The caller's `for t in range(T): env.step(actions[t])` loop (bin/interactive.py:27-36)
        with the caller's own actions, as ONE launch.  Outputs: outputs(g) per step, the state in world.pos / vel."""
        self.served_to += int(T)      # (ring() checks the commands against the launches: this one is about to start)
        self.ring(T)
        self.served_to -= int(T)
        self.start(T, on_current_stream=True)
        return self

    def start(self, T, on_current_stream=False):
        """Launch the server for global steps [served_to, served_to + T) on the server's stream, behind the current stream's
        work so far (the state it starts from) and behind the previous server launch."""
        cur = torch.cuda.current_stream(self.world.device)
        target = cur if on_current_stream else self.stream
        if not on_current_stream:
            self.stream.wait_stream(cur)
        b = self.blocks.bufs
        b.act = b.ids = b.u = None
        with torch.cuda.stream(target):
            ev = None
            if self.launch_events is not None:      # (bench.py: the server launch's own duration, HIP events on ITS stream)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), int(T))
                ev[0].record()
            _abi.check(self._L.mpe_step_server_start(C.byref(self._desc), C.byref(b), self.B, int(T), self.episode_len, self._lr,
                                                     self.seed, int(self.served_to), int(self.world.world_offset),
                                                     C.byref(self._srv), _abi.raw_stream(self.world.device)), "mpe_step_server_start")
            if ev is not None:
                ev[1].record()
                self.launch_events.append(ev)
            if self.comm is not None:      # the agents' comm state after the launch = what they said at its last step (core.py:171-177)
                self.env._comm.copy_(self.comm[(self.served_to + int(T) - 1) % int(self.comm.shape[0])])
        self.served_to += int(T)
        self.env._scenario_state_stale = True
        self.env._fast_acts.clear()

    def ring(self, n=1):
        """Command the next n steps: a one-thread launch on the CURRENT stream (behind whatever wrote their moves)."""
        if _abi.raw_stream(self.world.device).value != self._commander:
            raise _abi.MpeError("ring(): command from the stream that was current when the StepServer was built (the one its server "
                                "stream was probed against)")
        self.commanded += int(n)
        if self.commanded > self.served_to:
            raise _abi.MpeError("ring(): %d steps commanded, the launches started so far serve %d -- start() first" % (self.commanded, self.served_to))
        _abi.check(self._L.mpe_step_server_ring(C.byref(self._srv), int(n), _abi.raw_stream(self.world.device)), "mpe_step_server_ring")

    def wait(self, completed=None):
        """The current stream continues when `completed` steps (default: all commanded so far) have their outputs in memory."""
        _abi.check(self._L.mpe_step_server_wait(C.byref(self._srv), self.B, int(self.commanded if completed is None else completed),
                                                _abi.raw_stream(self.world.device)), "mpe_step_server_wait")

    def join(self):
        """The current stream waits for the END of the server launches started so far (all their steps were commanded)."""
        torch.cuda.current_stream(self.world.device).wait_stream(self.stream)

    def outputs(self, g):
        k = int(g) % self.slots
        return self.blocks.obs[k], self.blocks.rew[k], self.blocks.done[k]

    def check(self):
        """After a synchronisation: raise if the server or a wait gave up (a doorbell that never rang)."""
        st = int(self.status.item())
        if st:
            raise _abi.MpeError("step server: %s timed out waiting (status %d)" % ("the server" if st == 1 else "a wait", st))


class ServedRollout(object):
    """The random-action benchmark protocol through the step server: every step is COMMANDED on its own (a doorbell launch on the
    caller's stream, behind the launch that drew its moves), none is launched.  Per `episode_len` steps: ONE
    `mpe_random_actions_block` draw into one half of a 2 x episode_len move ring (the other half is being read), ONE server launch
    of episode_len steps on the server's stream (the episode's reset is the server's in-launch reset: mpe_reset's draws), and
    episode_len doorbells.  Fresh moves for every step, every step's rows / rewards / dones / state written -- the work of
    RandomRollout(regenerate=True).enqueue, bit for bit (tests/test_gpu_server.py).

    graphs=True: the caller-stream half of an episode -- the draw and its episode_len doorbells -- is ONE HIP graph per ring half,
    replayed every other episode (its draw repeats the moves of episodes 0 / 1: the graph protocol's usual frozen step numbers);
    the server launches stay eager on the server's own, PROBED stream (inside a graph the mapping of branches to hardware
    queues is not ours to probe).  ring_ahead=True: ONE doorbell per episode (all its steps commanded at once: their moves all
    exist after the draw) -- the server's own rate, where per-step doorbells measure the command processor's launch rate."""

    def __init__(self, env, episode_len=25, seed=None, slots=2, timeout_s=5.0, graphs=False, ring_ahead=False, max_launch_episodes=160,
                 launch_per_episode=False):
        w = env.world
        env._ensure_buffers()
        A, B = len(w.agents), w.batch_size
        self.env, self.world, self.A, self.B, self.EP = env, w, A, B, int(episode_len)
        assert self.EP >= 1
        self.moves = torch.zeros((2 * self.EP, A, B, _abi.MPE_ACTION_DIM), dtype=torch.float32, device=w.device)
        self.words, self.speakers = None, 0      # communication scenarios: the speakers' uniform random words, drawn with the moves
        if env._comm is not None:
            self.words = torch.zeros((2 * self.EP, A, B, int(w.dim_c)), dtype=torch.float32, device=w.device)
            for i, agent in enumerate(w.agents):
                if not agent.silent:
                    self.speakers |= 1 << i
        # graphs: ONE commanding stream of our own for capture, replay and the server-stream probe (a graph's kernels keep a tie to
        # the stream they were captured on; the default stream cannot capture)
        self.cmd = torch.cuda.Stream(device=w.device) if graphs else None
        # launch_per_episode: ONE server launch per episode, started BEHIND that episode's doorbell (ring_ahead) -- every command
        # precedes its launch, nothing in the launch ever waits, so the grid need not be resident: any batch size (1 048 576 worlds:
        # 16 384 workgroups).  The launches sit on the server's stream, the next episode's draw overlaps them on the commanding one;
        # a ring half is handed back by an event behind the launch that read it.
        self.launch_per_episode = bool(launch_per_episode)
        if self.launch_per_episode:
            ring_ahead = True
            self._read_done = [torch.cuda.Event(), torch.cuda.Event()]
            self._recorded = [False, False]
        if graphs:
            self.cmd.wait_stream(torch.cuda.current_stream(w.device))
            with torch.cuda.stream(self.cmd):
                self.srv = StepServer(env, self.moves, slots=slots, episode_len=self.EP, seed=seed, timeout_s=timeout_s, probe_graph=True,
                                      ahead=self.launch_per_episode, comm=self.words)
        else:
            self.srv = StepServer(env, self.moves, slots=slots, episode_len=self.EP, seed=seed, timeout_s=timeout_s,
                                  ahead=self.launch_per_episode, comm=self.words)
        self.seed = self.srv.seed
        self._L = _abi.lib()
        self.ring_ahead, self.max_launch = bool(ring_ahead), int(max_launch_episodes) * self.EP
        self.t = 0
        self._graphs = None
        if graphs:
            # Captured HERE, while no server is resident: torch's capture synchronises the device, and a resident server that is
            # waiting for doorbells nobody has rung yet would never let that return.  (Capture records, it does not run: the
            # doorbell word is untouched; the draw's code object is loaded by one eager call, the ring's by the probe.)
            with torch.cuda.stream(self.cmd):
                _abi.check(self._L.mpe_random_actions_block(self.moves.data_ptr(), None, A, B, self.seed, 0, 1, int(w.world_offset),
                                                            _abi.raw_stream(w.device)), "mpe_random_actions_block")
                gs = []
                for h in (0, 1):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.cmd):
                        self._caller_half(h, h * self.EP)
                    gs.append(g)
            self._graphs = gs

    def _caller_half(self, h, t):
        """The caller-stream work of the episode that starts at global step t and uses ring half h: draw, then the doorbells."""
        dev = self.world.device
        _abi.check(self._L.mpe_random_actions_block(self.moves[h * self.EP].data_ptr(), None, self.A, self.B, self.seed, int(t),
                                                    self.EP, int(self.world.world_offset), _abi.raw_stream(dev)), "mpe_random_actions_block")
        if self.words is not None:
            _abi.check(self._L.mpe_random_comm(self.words[h * self.EP].data_ptr(), self.A, self.B, int(self.world.dim_c), self.speakers,
                                               self.seed, int(t), self.EP, int(self.world.world_offset), _abi.raw_stream(dev)), "mpe_random_comm")
        if self.ring_ahead:      # the whole episode commanded at once (its moves all exist): the server's own rate
            _abi.check(self._L.mpe_step_server_ring(C.byref(self.srv._srv), self.EP, _abi.raw_stream(dev)), "mpe_step_server_ring")
            return
        for _k in range(self.EP):
            _abi.check(self._L.mpe_step_server_ring(C.byref(self.srv._srv), 1, _abi.raw_stream(dev)), "mpe_step_server_ring")

    def enqueue(self, steps):
        """`steps` (a multiple of episode_len) commanded steps; returns when everything is ENQUEUED, the current stream joined
        behind the server launch(es).  ONE server launch serves up to `max_launch` steps: nothing waits behind the resident
        kernel on its stream, and a ring half is handed back by the server's completion flags (a wait launch on the commanding
        stream in front of the draw that overwrites it), not by a launch boundary."""
        assert steps % self.EP == 0, "whole episodes"
        dev = self.world.device
        if self.cmd is not None and torch.cuda.current_stream(dev) != self.cmd:
            caller = torch.cuda.current_stream(dev)
            self.cmd.wait_stream(caller)
            with torch.cuda.stream(self.cmd):
                self.enqueue(steps)
            caller.wait_stream(self.cmd)
            return
        if self.launch_per_episode:
            cur = torch.cuda.current_stream(dev)
            for _ in range(steps // self.EP):
                h = (self.t // self.EP) & 1
                if self._recorded[h]:
                    cur.wait_event(self._read_done[h])       # the launch that read this half has ended
                self.srv.served_to += self.EP                 # (the commands come first: this episode's launch follows)
                if self._graphs is None:
                    self._caller_half(h, self.t)
                else:
                    self._graphs[h].replay()
                self.srv.commanded += self.EP
                self.srv.served_to -= self.EP
                self.srv.start(self.EP)                       # behind the doorbell just rung (start waits for the current stream)
                self._read_done[h].record(self.srv.stream)
                self._recorded[h] = True
                self.t += self.EP
            self.srv.join()
            return
        left = steps
        while left > 0:
            n = min(left, self.max_launch)
            self.srv.start(n)                           # (behind the current stream's work so far: the state it starts from)
            for _ in range(n // self.EP):
                h = (self.t // self.EP) & 1
                if self.t >= 2 * self.EP:
                    self.srv.wait(self.t - self.EP)     # the episode that read this half (steps t - 2 EP .. t - EP) is complete
                if self._graphs is None:
                    self._caller_half(h, self.t)
                else:
                    self._graphs[h].replay()
                self.srv.commanded += self.EP
                self.t += self.EP
            self.srv.join()
            left -= n


def step_many(env, moves, episode_len=0, seed=None, comm=None):
    """`for t in range(T): obs_n, reward_n, done_n, _ = env.step(moves[t])` as ONE launch: moves [T, A, B, 5] one-hot rows on the
    env's device (the reference's action format, environment.py:174-181, stacked over the steps).  -> a list of T tuples
    (obs_n, rew [A, B], done [A, B]) of views into the server's T output blocks (valid until the next call with the same T);
    world.pos / vel hold the state after the last step.  Bit-identical to the T env.step calls (tests/test_gpu_server.py).
    episode_len > 0: the worlds restart (world.reset_uniform's device draws) every episode_len steps, counted over the calls.
    comm: [T, A, B, dim_c] utterance rows for the communication scenarios."""
    T = int(moves.shape[0])
    key = (T, int(episode_len))
    cache = env.__dict__.setdefault("_step_many_servers", {})
    prog = getattr(env, "_prog", None)
    if prog is not None or _abi.lib().mpe_step_server_supported(C.byref(env._desc), env.world.batch_size) != 1:
        return _rollout_actions(env, moves, int(episode_len), seed, comm, cache, key)
    srv = cache.get(key)
    if srv is None or srv.moves.data_ptr() != moves.data_ptr() or tuple(srv.moves.shape) != tuple(moves.shape) or \
            (comm is not None and (srv.comm is None or srv.comm.data_ptr() != comm.data_ptr())):
        t0 = 0 if srv is None else srv.served_to
        srv = StepServer(env, moves, slots=T, episode_len=episode_len, seed=seed, probe=False, ahead=True, comm=comm)
        srv.served_to = srv.commanded = t0 - t0 % T      # (blocks and move tensors are indexed by the global step modulo T)
        if t0 % T:
            raise _abi.MpeError("step_many: a new move tensor mid-way through a block of %d steps" % T)
        srv.door.fill_(srv.commanded)
        srv.flag.fill_(srv.commanded)
        cache[key] = srv
    g0 = srv.served_to
    srv.run(T)
    return [srv.outputs(g0 + t) for t in range(T)]


def _rollout_actions(env, moves, episode_len, seed, comm, cache, key):
    """step_many for the envs the step server does not serve: row-program envs (user scenarios, traced reference-style files:
    mpe_rollout_rows_actions) and simple_spread / simple_tag beyond 16 entities (mpe_rollout_actions) -- their fused T-step rollout
    with the CALLER's moves (RollArgs.act_seq) instead of moves drawn in the kernel."""
    if not env.fused:
        raise _abi.MpeError("step_many: this env steps through Python callbacks (use env.step / GraphedStep)")
    if env._py_obs or env._py_reward or env._py_done or env._py_info:
        raise _abi.MpeError("step_many evaluates the device-side callbacks only")
    w = env.world
    env._ensure_buffers()
    A, B, T = len(w.agents), w.batch_size, int(moves.shape[0])
    if tuple(moves.shape[1:]) != (A, B, _abi.MPE_ACTION_DIM) or moves.dtype != torch.float32 or not moves.is_contiguous() \
            or moves.device != w.device:
        raise _abi.MpeError("moves: a contiguous float32 [T, A, B, %d] tensor on the env's device" % _abi.MPE_ACTION_DIM)
    prog = getattr(env, "_prog", None)
    if comm is not None or any(not a.silent for a in w.agents):
        raise _abi.MpeError("step_many: speaking agents are served for the built-in scenarios at their wave-per-agent shapes only")
    if prog is not None and bool(getattr(env, "_episode_in_launch", False)):
        raise _abi.MpeError("step_many: this env ends its episodes inside the launch (done programs / max_episode_steps): use env.step")
    if episode_len and not env._device_restart_ok:
        raise _abi.MpeError("step_many(episode_len > 0): the in-launch resets are world.reset_uniform's device draws; this env's reset_world is not that")
    st = cache.get(key)
    if st is None or not isinstance(st, dict):
        st = cache[key] = {"traj": Trajectory(env, T), "t": 0, "desc": _copy_struct(env._desc)}
    traj = st["traj"]
    b = traj.bufs
    b.act = b.ids = b.u = None
    L = _abi.lib()
    sd = int(w.seed if seed is None else seed) & (2 ** 64 - 1)
    lr = float(getattr(env._scenario, "landmark_range", 1.0))
    if prog is not None:
        _abi.check(L.mpe_rollout_rows_actions(C.byref(st["desc"]), C.byref(b), prog.ref, B, T, int(episode_len), lr, sd, int(st["t"]),
                                              int(w.world_offset), 1, moves.data_ptr(), _abi.raw_stream(w.device)), "mpe_rollout_rows_actions")
    else:
        _abi.check(L.mpe_rollout_actions(C.byref(st["desc"]), C.byref(b), B, T, int(episode_len), lr, sd, int(st["t"]),
                                         int(w.world_offset), 1, moves.data_ptr(), _abi.raw_stream(w.device)), "mpe_rollout_actions")
    st["t"] += T
    env._scenario_state_stale = True
    env._fast_acts.clear()
    return [(traj.obs[t], traj.rew[t], traj.done[t]) for t in range(T)]
