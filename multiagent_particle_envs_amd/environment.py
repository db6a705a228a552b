"""Batched MultiAgentEnv: the reference's Gym-style adapter (multiagent/environment.py:9-283) for
B worlds at once, with the body of step() replaced by ONE HIP kernel launch.

API kept from the reference (same names, same return structure):
    env.n, env.action_space[i], env.observation_space[i], env.world, env.agents,
    env.shared_reward, env.discrete_action_input, env.force_discrete_action
    obs_n = env.reset()
    obs_n, reward_n, done_n, info_n = env.step(action_n)      # lists of length n
With batch_size=B each list entry gains a leading batch axis: obs_n[i] is a [B, D_i] tensor,
reward_n[i] a [B] tensor, done_n[i] a [B] bool tensor (always False: environment.py:132-135),
info_n = {'n': [...]}.  In reference-compatibility mode (`batch_size=None` in make_env: one
world, NumPy in / NumPy out, NumPy-global-RNG resets) the env is used exactly like the reference's.

Two execution paths:
  fused    the nine built-in scenarios (unmodified callbacks): action decode, World.step and every agent's
           observation/reward/done/info in one `mpe_step` launch (csrc/mpe_split.hip wave-per-agent +
           reward wave for up to 6 agents, csrc/mpe_wide.hip wave-per-world for large N).
  generic  any user Scenario: actions decoded with tensor ops (all of _set_action's modes,
           environment.py:144-192), physics by `mpe_world_step`, then the user's Python
           reward/observation callbacks on [B, .] tensor views.
There is no CPU fallback on either path.

Output lifetime: step()/reset() return views of ping-pong device buffers -- the arrays returned by
call k stay intact until call k+2 returns (so `obs_n`/`new_obs_n` of the usual training loop never
alias).  Pass fresh_outputs=True for reference-style freshly allocated outputs on every call.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _abi, spaces


class _OutputSet(object):
    """One set of device output buffers + the ctypes MpeBuffers that points at them."""

    def __init__(self, env, obs=None):
        w, dev = env.world, env.world.device
        A, B = len(w.agents), w.batch_size
        off = env._obs_off
        self.obs = obs if obs is not None else torch.zeros(int(off[-1]) * B, dtype=torch.float32, device=dev)
        self.obs_n = [self.obs[off[i] * B: off[i + 1] * B].view(B, off[i + 1] - off[i]) for i in range(A)]
        self.rew = torch.zeros((A, B), dtype=torch.float32, device=dev)
        self.done = torch.zeros((A, B), dtype=torch.bool, device=dev)
        self.info = {}
        if env._benchmark:
            if env._kind == _abi.MPE_SCN_SPREAD:
                self.info = {"rew": torch.zeros((A, B), dtype=torch.float32, device=dev),
                             "collisions": torch.zeros((A, B), dtype=torch.int32, device=dev),
                             "min_dists": torch.zeros((A, B), dtype=torch.float32, device=dev),
                             "occupied_landmarks": torch.zeros((A, B), dtype=torch.int32, device=dev)}
            elif env._kind in (_abi.MPE_SCN_TAG, _abi.MPE_SCN_WORLD_COMM):
                self.info = {"collisions": torch.zeros((A, B), dtype=torch.int32, device=dev)}
            elif env._kind == _abi.MPE_SCN_ADVERSARY:   # squared distances: to the goal [A,B], to every landmark [L,A,B]
                self.info = {"goal_d2": torch.zeros((A, B), dtype=torch.float32, device=dev),
                             "landmark_d2": torch.zeros((len(w.landmarks), A, B), dtype=torch.float32, device=dev)}
        b = _abi.MpeBuffers()
        b.pos, b.vel = w.pos.data_ptr(), w.vel.data_ptr()
        b.obs, b.rew, b.done = self.obs.data_ptr(), self.rew.data_ptr(), self.done.data_ptr()
        if "rew" in self.info:
            b.info_rew = self.info["rew"].data_ptr()
            b.info_min_dists = self.info["min_dists"].data_ptr()
            b.info_occupied = self.info["occupied_landmarks"].data_ptr()
        if "collisions" in self.info:
            b.info_collisions = self.info["collisions"].data_ptr()
        if "goal_d2" in self.info:
            b.info_rew = self.info["goal_d2"].data_ptr()
            b.info_min_dists = self.info["landmark_d2"].data_ptr()
        if env._entity_table is not None:
            b.entity_table = env._entity_table.data_ptr()
        if w.choice_i32 is not None:      # per-world picks (goal landmark ...): updated in place by resets
            b.choice = w.choice_i32.data_ptr()
        if env._comm is not None:         # communication action rows = the agents' comm state
            b.comm = env._comm.data_ptr()
        self.bufs = b
        self.bufs_ref = C.byref(b)
        self.reward_n = [self.rew[i] for i in range(A)]
        self.done_n = [self.done[i] for i in range(A)]
        self.act_ptr = None       # MultiAgentEnv.step's fast path: the action pointer b.act currently holds

    def info_n(self, env):
        A = len(env.agents)
        if not env._benchmark:
            return {"n": [{} for _ in range(A)]}
        if env._kind == _abi.MPE_SCN_SPREAD:
            i_ = self.info
            return {"n": [(i_["rew"][i], i_["collisions"][i], i_["min_dists"][i], i_["occupied_landmarks"][i])
                          for i in range(A)]}
        if env._kind in (_abi.MPE_SCN_TAG, _abi.MPE_SCN_WORLD_COMM):
            return {"n": [self.info["collisions"][i] for i in range(A)]}
        if env._kind == _abi.MPE_SCN_ADVERSARY:     # simple_adversary.py:57-67: a scalar for an adversary, a tuple for a good agent
            g, l = self.info["goal_d2"], self.info["landmark_d2"]
            return {"n": [g[i] if a.adversary else tuple(l[k][i] for k in range(l.shape[0])) + (g[i],)
                          for i, a in enumerate(env.agents)]}
        return {"n": [{} for _ in range(A)]}


class MultiAgentEnv(object):
    metadata = {'render.modes': []}

    def __init__(self, world, reset_callback=None, reward_callback=None, observation_callback=None,
                 info_callback=None, done_callback=None, shared_viewer=True,
                 numpy_io=False, fresh_outputs=False, fused=None, max_episode_steps=None, auto_reset=False,
                 probe_placement=True, compile_program=None):
        self.world = world
        # a row program compiled in (compile_program()): None = attach an image lib/rows_cache/ already holds, never run
        # hipcc; True = compile whenever the program or its constants are new; False = always interpret
        self.compile_program_policy = compile_program
        self.probe_placement = bool(probe_placement)
        self.placement_probe = None
        self.agents = self.world.policy_agents
        self.n = len(world.policy_agents)
        self.batch_size = world.batch_size
        self.reset_callback = reset_callback
        self.reward_callback = reward_callback
        self.observation_callback = observation_callback
        self.info_callback = info_callback
        self.done_callback = done_callback
        # environment parameters (reference defaults, environment.py:28-36)
        self.discrete_action_space = True
        self.discrete_action_input = False
        self.force_discrete_action = world.discrete_action if hasattr(world, 'discrete_action') else False
        self._shared_reward = bool(world.collaborative) if hasattr(world, 'collaborative') else False
        self.time = 0
        # episode bookkeeping -- new API: the reference never ends an episode (environment.py:132-135,
        # done_callback is None in make_env.py:41-43) and leaves counting to the caller.  Off by default (parity).
        self.max_episode_steps = int(max_episode_steps) if max_episode_steps else 0
        self.auto_reset = bool(auto_reset)
        if self.auto_reset and not self.max_episode_steps:
            raise _abi.MpeError("auto_reset needs max_episode_steps")
        self.episode_step = None      # int32 [B] on the device: steps since each world's last reset
        self._steps_taken = 0         # env steps since construction
        self._may_finish = set()      # values of _steps_taken at which some world can reach the horizon
        self._horizon_clock_lost = False   # a device-side rollout advanced the per-world step counters behind the host's back
        self.numpy_io = bool(numpy_io)
        self.fresh_outputs = bool(fresh_outputs)
        self._step_impl = "split"
        self._constants_seen = world._constants_version
        self._scenario_state_stale = False   # set by device-side rollouts (RandomRollout): Scenario._apply pending

        # ---- can the whole step run as one fused kernel? ---------------------------------------------
        sc = getattr(observation_callback, "__self__", None)
        kind = getattr(sc, "kind", None)
        pkg = __name__.rsplit(".", 1)[0] + ".scenarios."
        builtin = next((c for c in type(sc).__mro__ if c.__module__.startswith(pkg)), None) if sc is not None else None
        def is_builtin(cb, name):
            return getattr(cb, "__self__", None) is sc and getattr(cb, "__func__", None) is builtin.__dict__.get(name)
        # the kernel side of a step (action decode, World.step, observation rows, reward) needs a built-in scenario's world,
        # a reset_world of the same scenario object, no scripted agents, no action / communication noise, and a kernel
        # for this shape; observation / reward / benchmark_data / done may each be the built-in (computed in the same
        # launch) or ANY Python callback (evaluated on the post-step world after the launch: "partial fusion").  A scenario
        # that overrides BOTH observation and reward has nothing left for the kernel but World.step: generic path.
        own = builtin is not None and kind is not None and \
            getattr(reset_callback, "__self__", None) is sc and getattr(observation_callback, "__self__", None) is sc and \
            (is_builtin(observation_callback, "observation") or is_builtin(reward_callback, "reward")) and \
            len(world.scripted_agents) == 0 and not any(l.movable for l in world.landmarks) and \
            all((a.silent or (kind in _abi.COMM_KINDS and not a.c_noise)) and not a.u_noise for a in world.agents)
        if own:   # ... and a kernel for this shape (mpe_split.hip's table: the reference's team sizes and some others)
            own = _abi.lib().mpe_step_supported(C.byref(world.scenario_desc(kind, getattr(sc, "num_adversaries", 0)))) == 1
        # the observation is a Python callback: the launch still writes the built-in rows (into the env's buffers, unused:
        # they are 4 of the kernel's 5-6 us at simple_spread's 65 536 worlds, against the ~150 launches of the generic path)
        self._py_obs = own and not is_builtin(observation_callback, "observation")
        self._py_reward = own and not is_builtin(reward_callback, "reward")
        self._py_info = own and info_callback is not None and not (
            is_builtin(info_callback, "benchmark_data") and kind in (_abi.MPE_SCN_SPREAD, _abi.MPE_SCN_TAG, _abi.MPE_SCN_ADVERSARY,
                                                                      _abi.MPE_SCN_WORLD_COMM))
        self._py_done = own and done_callback is not None
        # ---- ... or as World.step + an interpreted row program (rowspec.py: the scenario DESCRIBES its rows and rewards) ----
        # A scenario without a kernel of its own that supplies obs_spec / reward_spec for every agent steps in two launches
        # (mpe_world_step + mpe_rows) instead of the generic path's ~100; the specs win over Python observation / reward
        # callbacks of the same scenario (pass fused=False to keep the Python ones).
        self._prog = None
        self.two_launch_program = False
        self.finish_launch = True       # done_callback + auto_reset: restart finished worlds in one launch (mpe_episode_finish)
        if sc is None:
            sc = next((getattr(cb, "__self__", None) for cb in (reward_callback, reset_callback) if cb is not None), None)

        def own_or_absent(cb, name):
            # the scenario's OWN method of that name (whatever class of its MRO defines it), or no callback at all where the
            # scenario defines none either (a spec-only scenario: make_env passes None) -- never somebody else's function
            if cb is None:
                return getattr(sc, name, None) is None
            return getattr(cb, "__self__", None) is sc and getattr(cb, "__func__", None) is getattr(type(sc), name, None)
        if not own and fused is not False and sc is not None and \
                len(world.scripted_agents) == 0 and not any(l.movable for l in world.landmarks) and \
                len(world.entities) <= _abi.MPE_ROWS_MAX_ENTITIES and \
                not any(a.u_noise or (a.c_noise and not a.silent) for a in world.agents) and world.pos is not None:
            from . import rowspec
            if hasattr(sc, "obs_spec") and hasattr(sc, "reward_spec"):
                # the specs stand in for the scenario's own observation / reward -- and only for those: callbacks that are
                # not this scenario's methods (MultiAgentEnv(world, sc.reset_world, my_reward, my_obs)) are the caller's
                # rows and rewards and must run (generic path); fused=True with such callbacks is refused below
                if own_or_absent(observation_callback, "observation") and own_or_absent(reward_callback, "reward"):
                    self._prog = rowspec.compile_scenario(sc, world)
            elif builtin is not None and kind in rowspec.BUILTIN_PROGRAM_KINDS and is_builtin(observation_callback, "observation") and \
                    is_builtin(reward_callback, "reward") and getattr(reset_callback, "__self__", None) is sc:
                # a built-in scenario at a team size libmpe_hip.so has no kernel for: its callbacks are N-generic
                # (simple_adversary.py:69-139, simple_world_comm.py:126-289 loop over team lists) and so are their specs
                self._prog = rowspec.builtin_program(rowspec.BUILTIN_PROGRAM_KINDS[kind], world)
        if fused is None:
            fused = own or self._prog is not None
        if fused and not own and self._prog is None:
            raise _abi.MpeError("fused=True needs a built-in scenario with its own observation or its own reward callback at a "
                                "shape libmpe_hip.so has a kernel for (mpe_step_supported), or a scenario with obs_spec / "
                                "reward_spec, without scripted agents or noise")
        self.fused = bool(fused)
        if not self.fused or own:
            self._prog = None
        # Restarts drawn ON THE DEVICE (inside mpe_step_rows_episode / mpe_episode_finish / the episode rollouts) are
        # `world.reset_uniform(landmark_range, choices=choice_pops)` and nothing else: agents U[-1,1)^2, landmarks
        # U[-landmark_range, landmark_range)^2, vel = 0, uniform picks, utterances zeroed.  They never call reset_callback.
        # So they are taken only where that IS the scenario's reset_world: a built-in scenario's own reset_world, or a
        # scenario that says so (`device_reset = True`, scenario.py) -- otherwise finished worlds restart through the
        # masked reset_callback (a scenario with fixed posts, a restricted spawn area or per-world state of its own keeps
        # its distribution).  rng_mode 'numpy' (compatibility mode: the global np.random stream) never restarts on the device.
        rs = getattr(reset_callback, "__self__", None)
        rb = next((c for c in type(rs).__mro__ if c.__module__.startswith(pkg)), None) if rs is not None else None
        self._uniform_reset = rs is not None and (
            (rb is not None and getattr(reset_callback, "__func__", None) is rb.__dict__.get("reset_world")) or
            (bool(getattr(rs, "device_reset", False)) and getattr(reset_callback, "__func__", None) is getattr(type(rs), "reset_world", None)))
        self._has_speakers = any(not a.silent for a in world.agents)
        self._comm_kind = self.fused and (kind in _abi.COMM_KINDS if self._prog is None else self._has_speakers)
        self._scenario = sc
        self._kind = kind if self.fused and self._prog is None else _abi.MPE_SCN_GENERIC
        self._benchmark = self.fused and self._prog is None and info_callback is not None and not self._py_info
        if not self.fused:
            self._py_obs = self._py_reward = self._py_info = self._py_done = False
        if self._prog is not None:      # rows and rewards are the program's; benchmark_data / done stay Python callbacks
            self._py_obs = self._py_reward = False
            self._py_info = info_callback is not None
            self._py_done = done_callback is not None

        # ---- spaces (environment.py:38-70) -------------------------------------------------------------
        self.action_space = []
        self.observation_space = []
        if self._prog is not None:
            self._desc = self._program_desc()
            self._attach_program_image()
            self._obs_off = [int(self._desc.obs_off[i]) for i in range(len(world.agents) + 1)]
            obs_dims = list(self._prog.widths)
        elif self.fused:
            self._desc = world.scenario_desc(kind, getattr(sc, "num_adversaries", 0))
            self._obs_off = [int(self._desc.obs_off[i]) for i in range(len(world.agents) + 1)]
            obs_dims = [self._obs_off[i + 1] - self._obs_off[i] for i in range(len(world.agents))]
            if self._py_obs:    # the user's rows, whatever their width
                obs_dims = [int(observation_callback(agent, self.world).shape[-1]) for agent in self.agents]
        else:
            self._desc = None
            # (no observation callback: the reference's _get_obs returns an empty row, environment.py:120-123)
            obs_dims = [int(self._get_obs(agent).shape[-1]) for agent in self.agents]
        for agent, obs_dim in zip(self.agents, obs_dims):
            total_action_space = []
            if self.discrete_action_space:
                u_action_space = spaces.Discrete(world.dim_p * 2 + 1)
            else:
                u_action_space = spaces.Box(low=-agent.u_range, high=+agent.u_range, shape=(world.dim_p,),
                                            dtype=np.float32)
            if agent.movable:
                total_action_space.append(u_action_space)
            if self.discrete_action_space:
                c_action_space = spaces.Discrete(world.dim_c)
            else:
                c_action_space = spaces.Box(low=0.0, high=1.0, shape=(world.dim_c,), dtype=np.float32)
            if not agent.silent:
                total_action_space.append(c_action_space)
            if len(total_action_space) > 1:
                if all(isinstance(sp, spaces.Discrete) for sp in total_action_space):
                    act_space = spaces.MultiDiscrete([[0, sp.n - 1] for sp in total_action_space])
                else:
                    act_space = spaces.Tuple(total_action_space)
                self.action_space.append(act_space)
            else:
                self.action_space.append(total_action_space[0])
            self.observation_space.append(spaces.Box(low=-np.inf, high=+np.inf, shape=(obs_dim,), dtype=np.float32))
            agent.action.c = torch.zeros((self.batch_size, world.dim_c), dtype=torch.float32, device=world.device)

        # ---- device buffers for the fused path ----------------------------------------------------------
        self._sets = None
        self._flip = 0
        self._fast_acts = {}      # step()'s fast path: id -> weak reference of the caller's action tensors a full step validated
        self._act = None
        self._ids = None
        self._comm = None
        self._entity_table = None
        self.shared_viewer = shared_viewer

    def refresh_constants(self):
        """Re-read the entity / world constants (size, mass, collide, movable, max_speed, accel, dt, damping, contact
        force / margin) from the Python objects.  The reference reads them at every step; here they are snapshotted
        into the kernels' descriptor when the env is built, so call this after changing any of them on a live env
        (and rebuild RandomRollout / GraphedStep objects made from it)."""
        w = self.world
        w._desc = None
        w._entity_table = None
        self._constants_seen = w._constants_version
        # (a landmark made movable is integrated like any entity, core.py:158-169: physics through mpe_world_step, the
        #  scenario's callbacks in Python -- the fused output stages know agent velocities only)
        noisy = any(a.u_noise or (a.c_noise and not a.silent) for a in w.agents) or any(l.movable for l in w.landmarks)
        if self.fused and noisy:
            self.fused = False          # the fused kernels draw no action / communication noise (core.py:138,176): generic path
            self._comm_kind = False
            self._unfused_by_noise = True
        elif not self.fused and not noisy and getattr(self, "_unfused_by_noise", False):
            self.fused = True           # the noise is gone again: back to the single fused launch
            self._comm_kind = self._kind in _abi.COMM_KINDS
            self._unfused_by_noise = False
        if self.fused:
            self._desc = self._program_desc() if self._prog is not None else \
                w.scenario_desc(self._kind, getattr(self._scenario, "num_adversaries", 0))
            self._attach_program_image()
            if self._sets is not None:
                self._entity_table = w.entity_table(self._desc)
                self._desc_ref = C.byref(self._desc)
                for out in self._sets:
                    out.bufs.entity_table = self._entity_table.data_ptr()

    def _program_desc(self):
        """The descriptor of a row-program env: the world's physics constants (kind GENERIC) + the programs' row widths."""
        src = self.world.scenario_desc(_abi.MPE_SCN_GENERIC)
        d = _abi.MpeScenarioDesc()
        C.memmove(C.byref(d), C.byref(src), C.sizeof(d))
        off = 0
        for i, wd in enumerate(self._prog.widths):
            d.obs_off[i] = off
            off += int(wd)
        d.obs_off[len(self._prog.widths)] = off
        self._prog.validate(d)
        return d

    def _program_step(self, desc_ref, bufs_ref, B, st, probe=False):
        """World.step by the agents' waves, then every agent's row / reward / done from the post-step state by the row
        programs: ONE launch (`mpe_step_rows`; `two_launch_program = True` keeps mpe_world_step + mpe_rows, the A/B).
        With episodes that end on the device (`_episode_in_launch`) the same launch also counts the step, finds the finished
        worlds -- the program's done tests, the horizon -- and restarts them (`mpe_step_rows_episode`)."""
        L = _abi.lib()
        if self.two_launch_program:
            return L.mpe_world_step(desc_ref, bufs_ref, B, st) or L.mpe_rows(desc_ref, bufs_ref, self._prog.ref, B, st)
        if not probe and self._episode_in_launch:
            w = self.world
            if self.episode_step is None:
                self.episode_step = torch.zeros(self.batch_size, dtype=torch.int32, device=w.device)
                self._may_finish.add(self._steps_taken + self.max_episode_steps)
            # the episode number a restart draws with: one per step where a done test can fire at any step (as mpe_episode_finish
            # counts), one per step at which some world CAN reach the horizon otherwise (as the masked resets of that path count)
            # (after a device-side rollout drove this env -- RandomRollout counts its steps on the device only -- the host no longer
            #  knows at which steps a world can reach the horizon: every step then takes a fresh episode number, so that a restart
            #  never reuses a (seed, world, episode) key)
            counts = self._prog.has_done or self._horizon_clock_lost or (self._steps_taken + 1) in self._may_finish
            rc = L.mpe_step_rows_episode(desc_ref, bufs_ref, self._prog.ref, B, self.episode_step.data_ptr(), self.max_episode_steps,
                                         float(getattr(self._scenario, "landmark_range", 1.0)), int(w.seed) & (2 ** 64 - 1),
                                         int(w._episode), int(w.world_offset), st)
            if counts:
                w._episode += 1
                if hasattr(self._scenario, "_apply"):     # per-world Python state of the scenario (goal colours ...) follows lazily
                    self._scenario_state_stale = True
            return rc
        return L.mpe_step_rows(desc_ref, bufs_ref, self._prog.ref, B, st)

    @property
    def _episode_in_launch(self):
        """Episodes end INSIDE the step launch: a row-program env with max_episode_steps + auto_reset whose done condition is
        the program's (a `done_spec`) or the horizon alone -- no Python done callback, no Python rows."""
        return self._prog is not None and self.fused and self.auto_reset and bool(self.max_episode_steps) and self.finish_launch and \
            not self.two_launch_program and not self._py_done and not self._py_obs and not self._py_reward and self._device_restart_ok

    @property
    def _device_restart_ok(self):
        """Finished worlds may be restarted by the device-side draw (see `_uniform_reset` in __init__)."""
        return self._uniform_reset and self.world.rng_mode == "device"

    def _attach_program_image(self):
        """After the descriptor of a row-program env was (re)built: bring the compiled image in line with the policy."""
        prog = self._prog
        if prog is None or self.world.device.type != "cuda":
            return
        pol = self.compile_program_policy
        try:
            if pol is False:
                prog.unload()
            elif not prog.image_active(self._desc):
                prog.compile(self._desc, cached_only=pol is None)
        except _abi.MpeError:
            if pol:          # asked for explicitly: say why not
                raise
            # (None: a program that cannot be compiled in -- more than MPE_ROWS_STATIC_MAX_OPS ops -- runs interpreted)

    def compile_program(self, verbose=False):
        """Compile this env's row program IN (`mpe_rows_static_source` -> hipcc --genco -> `mpe_rows_load_image`; a few
        seconds the first time, cached by content under lib/rows_cache/): `step()` then launches the program as straight-line
        code specialised to this world's shape and constants -- bit-identical results, about the cost of a hand-fused kernel.
        The image is bound to the current constants: after an edit (an entity resized, dt changed) steps run interpreted again
        until the next compile_program().  Returns True when the image is the one the next step launches; raises MpeError for
        an env without a row program or one of more than 512 ops."""
        if self._prog is None:
            raise _abi.MpeError("compile_program: this env steps through %s, not through a row program"
                                % ("a fused kernel" if self.fused else "its scenario's torch callbacks"))
        if self.compile_program_policy is False:
            self.compile_program_policy = None
        self.refresh_constants()
        self._prog.compile(self._desc, verbose=verbose)
        return self._prog.image_active(self._desc)

    @property
    def program_compiled(self):
        """True when the next step launches a compiled image of the row program (False: interpreted, or no program)."""
        return self._prog is not None and self.fused and self._prog.image_active(self._desc)

    @property
    def step_impl(self):
        """Which kernel family the fused step launches: 'split' (wave-per-agent, `mpe_step`, the default) or
        'thread' (thread-per-world, `mpe_step_thread`: the independent second implementation the parity tests
        compare against -- bit-identical results)."""
        return self._step_impl

    @step_impl.setter
    def step_impl(self, value):
        if value not in ("split", "thread"):
            raise _abi.MpeError("step_impl is 'split' or 'thread'")
        self._step_impl = value
        if self._sets is not None and self._prog is None:
            self._mpe_step = _abi.lib().mpe_step_thread if value == "thread" else _abi.lib().mpe_step

    @property
    def shared_reward(self):
        """environment.py:36: every agent receives the sum of all rewards.  Read at every step by the reference, so it
        may be flipped on a live env: the fused kernels implement the scenario's own setting (world.collaborative), any
        other value sends the env to the generic path."""
        return self._shared_reward

    @shared_reward.setter
    def shared_reward(self, value):
        value = bool(value)
        if value != self._shared_reward and getattr(self, "fused", False):
            self.fused = False
            self._comm_kind = False
        self._shared_reward = value

    # ------------------------------------------------------------------------------------------
    def _ensure_buffers(self):
        if self._sets is not None:
            return
        w = self.world
        w._require_device()
        A, B = len(w.agents), w.batch_size
        self._entity_table = w.entity_table(self._desc)   # read by the wave-per-world (large N) kernel only
        self._desc_ref = C.byref(self._desc)
        self._mpe_step = _abi.lib().mpe_step_thread if self.step_impl == "thread" else _abi.lib().mpe_step
        if self._prog is not None:
            self._mpe_step = self._program_step
        if self._kind in _abi.COMM_KINDS or (self._prog is not None and w.dim_c > 0):
            self._comm = torch.zeros((A, B, w.dim_c), dtype=torch.float32, device=w.device)
            self._speak_mask = torch.tensor([0.0 if a.silent else 1.0 for a in w.agents], dtype=torch.float32,
                                            device=w.device).reshape(A, 1, 1)
        self._act = torch.zeros((A, B, _abi.MPE_ACTION_DIM), dtype=torch.float32, device=w.device)
        self._ids = torch.zeros((A, B), dtype=torch.int32, device=w.device)
        self._sets = [_OutputSet(self, o) for o in self._place_observation_buffers(2)]

    # observation blocks from this size on are placed by a timed probe (below)
    PROBE_MIN_BYTES = 128 << 20
    PROBE_CANDIDATES = 16

    def _place_observation_buffers(self, n):
        """Observation buffers for the env's `n` output sets.  Small ones are plain allocations.  For LARGE row blocks
        (simple_spread N=64 at 4096 worlds: 403 MB per set) the rate at which the step streams its rows is a stable
        property of the ALLOCATION -- 67-68 us per launch into some buffers, 82-84 us into others of the same size, in the
        same process, whatever the offset inside the buffer or the bits of its virtual address (profiles/
        r3_c4_placement_*.txt; the address-translation counters are flat, the L2's memory-side write stalls are DRAM-credit
        stalls: where the driver put the pages in HBM).  So a handful of candidates is allocated, the real step kernel is
        timed on each (state and outputs in scratch copies: the world does not move) until `n` of them take the rows at
        < 1.2 x the time of a plain fill of the block (at most PROBE_CANDIDATES), and the fastest `n` are kept.  Costs a few
        milliseconds and, transiently, that many blocks (peak: PROBE_CANDIDATES x the block, 6.4 GB for N = 64 at 4096 worlds;
        the rejected ones are returned to the device with empty_cache()), once per env; results do not depend on it."""
        w = self.world
        nfl = int(self._obs_off[-1]) * w.batch_size
        # (only the wave-per-world kernels -- more than 16 entities, dozens of agent blocks written side by side -- are
        #  placement-sensitive; the 1M-world N=3 step writes three contiguous 75 MB blocks and is not: 68-70 us anywhere)
        # (not for an env that allocates fresh outputs on every call -- it never uses the placed buffers -- and not under
        #  stream capture: the probe synchronises on events)
        if not self.probe_placement or not self.fused or nfl * 4 < self.PROBE_MIN_BYTES or len(w.entities) <= 16 or \
                self._kind not in (_abi.MPE_SCN_SPREAD, _abi.MPE_SCN_TAG) or self.fresh_outputs or \
                torch.cuda.is_current_stream_capturing():
            return [None] * n
        dev = w.device
        first = torch.zeros(nfl, dtype=torch.float32, device=dev)
        scratch = _OutputSet(self, first)    # rew / done / info scratch of the probe launches; its obs pointer is swapped below
        b = scratch.bufs
        pos, vel = w.pos.clone(), w._vel_all.clone()
        b.pos, b.vel = pos.data_ptr(), vel.data_ptr()
        b.act, b.ids, b.u = None, self._ids.data_ptr(), None
        st, L = self._stream(), _abi.lib()
        step = (lambda *a: self._program_step(*a, probe=True)) if self._prog is not None else self._mpe_step

        def timed(t, launches=8):
            b.obs = t.data_ptr()
            for _ in range(2):
                _abi.check(step(self._desc_ref, scratch.bufs_ref, w.batch_size, st), "mpe_step (placement probe)")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                step(self._desc_ref, scratch.bufs_ref, w.batch_size, st)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / launches
        # a plain fill of the block is placement-insensitive: buffers the step streams into at < 1.2 x that are "fast" ones
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        first.fill_(0.0)
        e0.record()
        for _ in range(4):
            first.fill_(0.0)
        e1.record()
        e1.synchronize()
        fill_ms = e0.elapsed_time(e1) / 4
        cands = [(timed(first), 0, first)]
        try:
            while len(cands) < self.PROBE_CANDIDATES and sum(1 for c in cands if c[0] < 1.2 * fill_ms) < n:
                t = torch.zeros(nfl, dtype=torch.float32, device=dev)
                cands.append((timed(t), len(cands), t))
        except torch.cuda.OutOfMemoryError:      # keep what fits
            pass
        cands.sort(key=lambda c: c[0])
        self.placement_probe = {"candidates_ms": [round(c[0], 5) for c in cands], "kept": n, "fill_ms": round(fill_ms, 5)}
        keep = [c[2] for c in cands[:n]]
        while len(keep) < n:
            keep.append(None)
        for k in keep:
            if k is not None:
                k.zero_()
        # the losers go back to the DEVICE, not just to torch's caching allocator (up to 16 blocks of 403 MB at N = 64,
        # 4096 worlds: 6.4 GB peak, transient) -- the kept blocks are live tensors and stay where they are
        del cands, first, scratch, pos, vel, b, timed
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()
        return keep

    def _stream(self):
        return _abi.raw_stream(self.world.device)

    def _next_set(self):
        if self.fresh_outputs:
            return _OutputSet(self)
        self._flip ^= 1
        return self._sets[self._flip]

    def _stage_actions(self, action_n):
        """Bring the caller's actions into the [A,B,5] fp32 (or [A,B] int32) device layout; a tensor
        that already has it is used in place (zero copy)."""
        A, B = len(self.agents), self.batch_size
        if self._comm is not None and self._has_speakers:
            if isinstance(action_n, tuple):
                # (moves [A,B,5], utterances [A,B,dim_c]) as two device tensors -- the batched form of the reference's per-agent
                # [move | utterance] rows: the moves are used in place, the utterances land in the comm state with ONE launch (rows
                # of silent agents are masked to zero) instead of two small copies per agent
                dc = self.world.dim_c
                if len(action_n) != 2 or not all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                                                 for t in action_n) or \
                        action_n[0].shape != (A, B, _abi.MPE_ACTION_DIM) or action_n[1].shape != (A, B, dc):
                    raise _abi.MpeError("a tuple action is (moves [A,B,5], utterances [A,B,dim_c]): two contiguous float32 device tensors "
                                        "(A = %d, B = %d, dim_c = %d)" % (A, B, dc))
                torch.mul(action_n[1], self._speak_mask, out=self._comm)
                act = action_n[0]
                if self.force_discrete_action:
                    act = torch.zeros_like(act).scatter_(-1, act.argmax(dim=-1, keepdim=True), 1.0)
                return act, None
            return self._stage_comm_actions(action_n), None
        if self.discrete_action_input:
            if torch.is_tensor(action_n) and action_n.shape == (A, B) and action_n.dtype == torch.int32 \
                    and action_n.is_cuda and action_n.is_contiguous():
                return None, action_n
            for i in range(A):
                self._ids[i].copy_(torch.as_tensor(action_n[i]).reshape(-1).expand(B) if not torch.is_tensor(
                    action_n[i]) else action_n[i].reshape(-1).expand(B))
            return None, self._ids
        if torch.is_tensor(action_n) and action_n.shape == (A, B, _abi.MPE_ACTION_DIM) \
                and action_n.dtype == torch.float32 and action_n.is_cuda and action_n.is_contiguous():
            act = action_n
        else:
            for i in range(A):
                a = action_n[i]
                a = a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a), dtype=torch.float32)
                self._act[i].copy_(a.reshape(-1, _abi.MPE_ACTION_DIM).expand(B, _abi.MPE_ACTION_DIM))
            act = self._act
        if self.force_discrete_action:  # environment.py:169-172: argmax -> one-hot
            idx = act.argmax(dim=-1, keepdim=True)
            act = torch.zeros_like(act).scatter_(-1, idx, 1.0)
        return act, None

    @staticmethod
    def _rows_of_one_tensor(action_n, A, B):
        """The reference's per-agent list, when it is the rows of ONE contiguous float32 [A,B,5] device tensor (a stacked policy
        output handed over as `[t[i] for i in range(A)]`): that tensor -- no copies -- else None."""
        if type(action_n) is not list or len(action_n) != A or not torch.is_tensor(action_n[0]):
            return None
        base = action_n[0]._base
        if base is None or base.shape != (A, B, _abi.MPE_ACTION_DIM) or base.dtype != torch.float32 or not base.is_cuda or \
                not base.is_contiguous():
            return None
        off, row = base.storage_offset(), B * _abi.MPE_ACTION_DIM
        for i, a in enumerate(action_n):
            if not torch.is_tensor(a) or a._base is not base or a.storage_offset() != off + i * row or \
                    a.shape != (B, _abi.MPE_ACTION_DIM) or not a.is_contiguous():
                return None
        return base

    def _stage_comm_actions(self, action_n):
        """Communication scenarios: agent i's action row is [move (5) if movable] + [utterance (dim_c) if not
        silent] (Discrete / MultiDiscrete spaces, environment.py:148-155, 183-190).  The move part goes to
        act [A,B,5], the utterance to comm [A,B,dim_c], which IS the agents' comm state after the step
        (update_agent_state, core.py:171-177)."""
        B, dc = self.batch_size, self.world.dim_c
        for i, agent in enumerate(self.agents):
            a = action_n[i]
            a = a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a), dtype=torch.float32)
            a = a.to(self._act.device, torch.float32).reshape(-1, a.shape[-1]).expand(B, a.shape[-1])
            k = 0
            if agent.movable:
                m = a[:, :_abi.MPE_ACTION_DIM]
                if self.force_discrete_action:   # environment.py:169-172: argmax -> one-hot
                    m = torch.zeros_like(m).scatter_(-1, m.argmax(dim=-1, keepdim=True), 1.0)
                self._act[i].copy_(m)
                k = _abi.MPE_ACTION_DIM
            if not agent.silent:
                self._comm[i].copy_(a[:, k:k + dc])
                k += dc
            assert k == a.shape[-1], "action row of agent %d has %d entries, expected %d" % (i, a.shape[-1], k)
        return self._act

    # ------------------------------------------------------------------------------------------
    def sync_from_device(self):
        """Re-derive the scenario's Python-side per-world state (goal-dependent colours, `agent.goal_a` views, comm
        state, keys ...) from the device tensors.  Device-side rollouts (`RandomRollout`, `mpe_rollout_random`) reset
        worlds with `mpe_reset` / in-kernel resets without going through `Scenario.reset_world`; they flag the env and
        this runs before the next `step()` / `reset()` / `benchmark_data` through the Python API."""
        self._scenario_state_stale = False
        sc, w = self._scenario if self._scenario is not None else getattr(self, "scenario", None), self.world
        if sc is not None and hasattr(sc, "_apply"):
            sc._apply(w)
        for i, agent in enumerate(w.agents):   # update_agent_state: silent agents 0, the others their last words
            if self._comm is not None and not agent.silent:
                agent.state.c = self._comm[i]
            else:
                agent.state.c = torch.zeros((self.batch_size, w.dim_c), dtype=torch.float32, device=w.device)

    def step_many(self, moves, episode_len=0, comm=None):
        """`for t in range(T): env.step(moves[t])` as ONE launch (new API; rollout.step_many, DESIGN.md 2 "the step server"):
        moves [T, n, B, 5] one-hot rows on the env's device -> [(obs_n, rew [n, B], done [n, B]) per step], bit-identical to the
        T step() calls; for callers whose actions exist ahead of the steps (recorded / scripted sequences, action repeat).
        The built-in scenarios (comm: [T, n, B, dim_c] utterance rows where agents speak), batched mode."""
        from .rollout import step_many
        return step_many(self, moves, episode_len, comm=comm)

    def step(self, action_n):
        """environment.py:80-104 for B worlds."""
        # ---- the common case in as few Python operations as it takes: one of the caller's preallocated [A,B,5] device
        # tensors again (one that an earlier step validated; contents rewritten in place), a fully fused non-communication
        # env, nothing flipped since.  One step() then costs the host ~3 us instead of ~7 (the headline kernel takes 5.5: the
        # Python API stops being host-bound).  The tensors are remembered by weak reference: nothing is kept alive.
        if type(action_n) is list and not self._comm_kind:      # the reference's per-agent list ... as the rows of ONE [A,B,5] device
            base = self._rows_of_one_tensor(action_n, len(self.agents), self.batch_size)      # tensor (a stacked policy output): that
            if base is not None:                                                           # tensor, no copies -- and the short path
                action_n = base
        pair = type(action_n) is tuple and len(action_n) == 2      # (moves, utterances) of a communication scenario
        known = self._fast_acts.get(id(action_n[0]) if pair else id(action_n))
        if known is not None and (type(known) is tuple) == pair and \
                (known[0]() is action_n[0] and known[1]() is action_n[1] if pair else known() is action_n) and \
                self._constants_seen == self.world._constants_version and \
                not self._scenario_state_stale and not self.discrete_action_input and self.discrete_action_space and \
                not self.force_discrete_action and len(self.agents) == len(self.world.agents) and self.fused and \
                not self.fresh_outputs:
            self._flip ^= 1
            out = self._sets[self._flip]
            if pair:      # one staging launch: the utterances into the comm state (silent agents' rows masked to zero)
                torch.mul(action_n[1], self._speak_mask, out=self._comm)
                for i, agent in enumerate(self.world.agents):
                    if not agent.silent:
                        agent.state.c = self._comm[i]
                action_n = action_n[0]
            p = action_n.data_ptr()
            if out.act_ptr != p:
                b = out.bufs
                b.act, b.ids, b.u = p, None, None
                out.act_ptr = p
            rc = self._mpe_step(self._desc_ref, out.bufs_ref, self.batch_size, self._stream())
            if rc:
                _abi.check(rc, "mpe_step")
            return list(out.obs_n), list(out.reward_n), list(out.done_n), out.info_n(self)
        if self._scenario_state_stale:
            self.sync_from_device()
        if len(self.agents) != len(self.world.agents):   # environment.py:85 re-reads world.policy_agents every step
            self.agents = self.world.policy_agents
        if self._constants_seen != self.world._constants_version:   # an entity / world constant was assigned: re-snapshot
            self.refresh_constants()
        if not self.fused or (self._comm_kind and self.discrete_action_input) or not self.discrete_action_space:
            return self._step_generic(action_n)      # (the kernels decode 5-wide move rows / integer ids only)
        self._ensure_buffers()
        act, ids = self._stage_actions(action_n)
        out = self._next_set()
        b = out.bufs
        b.act = act.data_ptr() if act is not None else None
        b.ids = ids.data_ptr() if ids is not None else None
        b.u = None
        for o in self._sets:    # (the fast path's note of what each set's b.act holds: re-established there.  EVERY set: whatever
            o.act_ptr = None    #  sent this call down the slow path -- a rollout that edited the bufs, a flag flipped -- may have touched both)
        # arm the fast path for the next call with this very tensor: zero-copy moves, nothing evaluated in Python afterwards
        if act is action_n and not self._comm_kind and not self.fresh_outputs and not self.numpy_io and \
                not self.max_episode_steps and not (self._py_obs or self._py_reward or self._py_done or self._py_info):
            if len(self._fast_acts) >= 64:
                self._fast_acts.clear()
            self._fast_acts[id(action_n)] = weakref.ref(action_n)
        elif pair and self._comm_kind and act is action_n[0] and not self.fresh_outputs and not self.numpy_io and \
                not self.max_episode_steps and not (self._py_obs or self._py_reward or self._py_done or self._py_info):
            if len(self._fast_acts) >= 64:
                self._fast_acts.clear()
            self._fast_acts[id(action_n[0])] = (weakref.ref(action_n[0]), weakref.ref(action_n[1]))
        if self._py_reward:          # the reward is a Python callback: the launch skips its reward stage
            b.rew = None
        rc = self._mpe_step(self._desc_ref, out.bufs_ref, self.batch_size, self._stream())
        if rc:
            _abi.check(rc, "mpe_step")
        if self._comm is not None:   # update_agent_state (core.py:171-177): state.c = action.c for the agents that speak
            for i, agent in enumerate(self.world.agents):
                if not agent.silent:
                    agent.state.c = self._comm[i]
        reward_n, done_n, info_n, done = out.reward_n, out.done_n, None, out.done
        if self._py_reward or self._py_done or self._py_info:
            # partial fusion: the user's reward / done / benchmark_data callbacks on the post-step world (environment.py:92-102)
            if self._py_reward:
                b.rew = out.rew.data_ptr()
                reward_n = [torch.as_tensor(self._get_reward(a), device=self.world.device) for a in self.agents]
                if self.shared_reward:
                    total = torch.stack([r.expand(self.batch_size) for r in reward_n]).sum(dim=0)
                    reward_n = [total] * self.n
            if self._py_done:
                done = torch.stack([torch.as_tensor(self._get_done(a), device=self.world.device).bool().expand(self.batch_size)
                                    for a in self.agents]).contiguous()
                done_n = [done[i] for i in range(self.n)]
            if self._py_info:
                info_n = {'n': [self._get_info(a) for a in self.agents]}
        if info_n is None:
            info_n = out.info_n(self)
        if self.max_episode_steps and self._episode_tick(done, out) and not self._py_obs:
            self._observe_into(out)         # worlds that finished were reset: their rows are the new episode's first
        obs_n = [self._get_obs(a) for a in self.agents] if self._py_obs else out.obs_n
        return self._deliver(obs_n, reward_n, done_n, info_n)

    _BUILTIN_NAMES = {_abi.MPE_SCN_SIMPLE: "simple", _abi.MPE_SCN_SPREAD: "simple_spread", _abi.MPE_SCN_TAG: "simple_tag",
                      _abi.MPE_SCN_ADVERSARY: "simple_adversary", _abi.MPE_SCN_PUSH: "simple_push",
                      _abi.MPE_SCN_SPEAKER_LISTENER: "simple_speaker_listener", _abi.MPE_SCN_REFERENCE: "simple_reference",
                      _abi.MPE_SCN_CRYPTO: "simple_crypto", _abi.MPE_SCN_WORLD_COMM: "simple_world_comm"}

    def _finish_program(self):
        """The row program mpe_episode_finish rewrites the restarted worlds' rows with: the env's own, or -- a built-in
        scenario stepping through a kernel of its own -- that scenario's rows as a program (bit-identical to the kernel's,
        tests/test_rowspec.py); None where there is none (more than 64 entities, Python observation rows, `finish_launch`
        switched off)."""
        if not self.finish_launch or not self.fused or self._py_obs or not self._device_restart_ok:
            return None
        if self._prog is not None:
            return self._prog
        if self.__dict__.get("_finish_prog") is None:
            self._finish_prog = False
            name = self._BUILTIN_NAMES.get(self._kind)
            if name is not None and len(self.world.entities) <= _abi.MPE_ROWS_MAX_ENTITIES and \
                    not any(l.movable for l in self.world.landmarks):
                from . import rowspec
                try:        # (a shape the built-in specs do not cover, regions with A > 32 ...: the mask-reset path, from the first step on)
                    prog = rowspec.builtin_program(name, self.world)
                    if list(prog.widths) == [self._obs_off[i + 1] - self._obs_off[i] for i in range(len(prog.widths))]:
                        prog.validate(self._desc)
                        if self.compile_program_policy is not False and self.world.device.type == "cuda":
                            try:
                                prog.compile(self._desc, cached_only=self.compile_program_policy is None)
                            except _abi.MpeError:
                                pass
                        self._finish_prog = prog
                except _abi.MpeError:
                    self._finish_prog = False
        return self._finish_prog or None

    def _episode_tick(self, done, out=None):
        """After a step: count it for every world, mark the worlds that reached max_episode_steps done (all agents),
        and -- auto_reset -- start their next episode (reset_callback with mask = the done row).  `done` is the
        [A,B] bool tensor the step wrote.  Returns True when a reset was issued (observations must be refreshed).
        No host synchronisation: which worlds finish is decided on the device; the host only tracks at which step
        counts some world CAN finish (max_episode_steps after every full, masked or automatic reset) to skip the
        reset launches everywhere else."""
        w = self.world
        if self._episode_in_launch:      # counted, decided and restarted by the step's own launch (mpe_step_rows_episode)
            self._steps_taken += 1
            if self._steps_taken in self._may_finish:
                self._may_finish.discard(self._steps_taken)
                self._may_finish.add(self._steps_taken + self.max_episode_steps)
            return False
        if self.episode_step is None:
            self.episode_step = torch.zeros(self.batch_size, dtype=torch.int32, device=w.device)
            self._may_finish.add(self._steps_taken + self.max_episode_steps)
        self._steps_taken += 1
        # a world can end at ANY step: a Python done_callback, or the program's own done tests where the restart cannot happen in
        # the step launch (a reset_world that is not the device-side draw: `_device_restart_ok`)
        ends_any_step = self.done_callback is not None or (self._prog is not None and self._prog.has_done)
        if ends_any_step and self.auto_reset and out is not None and self._finish_program() is not None:
            # a done_callback can end a world at ANY step.  ONE launch (mpe_episode_finish) counts the step, finds the worlds
            # some agent's done row (or the horizon) flagged, restarts exactly those -- reset_world's draws, counters, comm
            # state -- and rewrites their observation rows; a workgroup of 64 worlds none of which finished returns after
            # reading its flags, so a step at which nothing ends costs one small launch more (round 3: a tick, a mask
            # reduction, a masked reset, a comm fill and a full mpe_observe relaunch on every step).
            prog = self._finish_program()
            b = _abi.MpeBuffers()
            C.memmove(C.byref(b), C.byref(out.bufs), C.sizeof(b))
            b.done = done.data_ptr()
            b.act = b.ids = b.u = None
            _abi.check(_abi.lib().mpe_episode_finish(C.byref(self._desc), C.byref(b), prog.ref, self.batch_size,
                                                     self.episode_step.data_ptr(), self.max_episode_steps,
                                                     float(getattr(self._scenario, "landmark_range", 1.0)),
                                                     int(w.seed) & (2 ** 64 - 1), int(w._episode), int(w.world_offset),
                                                     self._stream()), "mpe_episode_finish")
            w._episode += 1
            if hasattr(self._scenario, "_apply"):     # per-world Python state of the scenario (goal colours ...) follows lazily
                self._scenario_state_stale = True
            return False          # (the rows of the restarted worlds are already the new episode's first)
        _abi.check(_abi.lib().mpe_episode_tick(self.episode_step.data_ptr(), done.data_ptr(), done.shape[0],
                                               self.batch_size, self.max_episode_steps, 1 if self.auto_reset else 0,
                                               self._stream()), "mpe_episode_tick")
        if ends_any_step and self.auto_reset:
            # (shapes without a row program -- more than 64 entities --, the generic path, and scenarios whose reset_world is
            #  their own -- `device_reset` not declared: the same in separate launches, the restart through reset_callback)
            # a done_callback can end a world at ANY step: restart every world some agent (or the horizon) flagged,
            # at every step, and restart its step counter too (the tick only clears it at the horizon).
            # Cost, knowingly paid: a masked reset_callback, a comm fill and (the caller's) mpe_observe relaunch on EVERY
            # step, whether or not a world finished -- skipping them when nothing finished would need the host to read a
            # device flag (a synchronisation per step, which costs more than the three small launches it saves); the
            # horizon-only path below knows on the host when a world CAN finish and skips them everywhere else.
            finished = done.any(dim=0)
            self.episode_step.masked_fill_(finished, 0)
            self.reset_callback(w, mask=finished)
            if self._comm is not None:
                self._comm.masked_fill_(finished[None, :, None], 0.0)
            return True
        if self._steps_taken not in self._may_finish:
            return False
        self._may_finish.discard(self._steps_taken)
        if not self.auto_reset:
            return False
        self.reset_callback(w, mask=done[0])
        if self._comm is not None:
            self._comm.masked_fill_(done[0][None, :, None], 0.0)
        self._may_finish.add(self._steps_taken + self.max_episode_steps)
        return True

    def _note_reset(self, mask):
        if not self.max_episode_steps:
            return
        if self.episode_step is not None:
            if mask is None:
                self.episode_step.zero_()
            else:
                self.episode_step.masked_fill_(torch.as_tensor(mask, device=self.world.device).bool(), 0)
        if mask is None:
            self._may_finish.clear()
            self._horizon_clock_lost = False     # every world starts an episode now: the host's clock is right again
        self._may_finish.add(self._steps_taken + self.max_episode_steps)

    def _observe_into(self, out):
        """Rewrite the observation rows of `out` from the current state (mpe_observe); rewards, dones, info stay."""
        b = out.bufs
        saved = (b.rew, b.done, b.info_rew, b.info_collisions, b.info_min_dists, b.info_occupied, b.act, b.ids, b.u)
        b.rew = b.done = b.info_rew = b.info_collisions = b.info_min_dists = b.info_occupied = None
        b.act = b.ids = b.u = None
        try:
            if self._prog is not None:
                _abi.check(_abi.lib().mpe_rows(C.byref(self._desc), C.byref(b), self._prog.ref, self.batch_size, self._stream()),
                           "mpe_rows")
            else:
                _abi.check(_abi.lib().mpe_observe(C.byref(self._desc), C.byref(b), self.batch_size, self._stream()),
                           "mpe_observe")
        finally:
            (b.rew, b.done, b.info_rew, b.info_collisions, b.info_min_dists, b.info_occupied, b.act, b.ids, b.u) = saved

    def reset(self, seeds=None, mask=None):
        """environment.py:106-116.  `seeds` (one per world) gives reference-exact initial states
        (`np.random.seed(s); env.reset()` per world, drawn on the host); `mask` resets a subset."""
        if self._scenario_state_stale:
            self.sync_from_device()
        world = self.world
        kw = {}
        if seeds is not None:
            kw["seeds"] = seeds
        if mask is not None:
            kw["mask"] = mask
        self.reset_callback(world, **kw)
        self._note_reset(mask)
        self.agents = world.policy_agents
        if not self.fused:
            obs_n = [self._get_obs(agent) for agent in self.agents]
            return self._deliver(obs_n, None, None, None)[0]
        self._ensure_buffers()
        if self._comm is not None:   # every reset_world zeroes the comm state (of the worlds it resets)
            if mask is None:
                self._comm.zero_()
            else:
                self._comm[:, torch.as_tensor(mask, device=self._comm.device).bool()] = 0.0
        if self._py_obs:
            return self._deliver([self._get_obs(agent) for agent in self.agents], None, None, None)[0]
        out = self._next_set()
        self._observe_into(out)
        return self._deliver(out.obs_n, None, None, None)[0]

    def _deliver(self, obs_n, reward_n, done_n, info_n):
        if not self.numpy_io:
            return list(obs_n), (list(reward_n) if reward_n is not None else None), \
                (list(done_n) if done_n is not None else None), info_n
        # reference-compatible single-world view: 1-D float arrays, Python floats/bools
        obs = [o[0].detach().cpu().numpy().astype(np.float64) for o in obs_n]
        rew = [float(r[0]) for r in reward_n] if reward_n is not None else None
        done = [bool(d[0]) for d in done_n] if done_n is not None else None
        info = None
        if info_n is not None:
            def conv(x):
                if isinstance(x, tuple):
                    return tuple(conv(y) for y in x)
                if isinstance(x, list):      # per-world Python objects (a reference-style benchmark_data): world 0's
                    return x[0] if x else x
                if torch.is_tensor(x):
                    v = x[0].item()
                    return v
                return x
            info = {"n": [conv(x) for x in info_n["n"]]}
        return obs, rew, done, info

    # ---- generic path: user callbacks + HIP physics ------------------------------------------------
    def _get_info(self, agent):
        if self.info_callback is None:
            return {}
        return self.info_callback(agent, self.world)

    def _get_obs(self, agent):
        if self.observation_callback is None:
            return torch.zeros((self.batch_size, 0), device=self.world.device)
        return self.observation_callback(agent, self.world)

    def _get_done(self, agent):
        if self.done_callback is None:
            return torch.zeros(self.batch_size, dtype=torch.bool, device=self.world.device)
        return self.done_callback(agent, self.world)

    def _get_reward(self, agent):
        if self.reward_callback is None:
            return torch.zeros(self.batch_size, dtype=torch.float32, device=self.world.device)
        return self.reward_callback(agent, self.world)

    def _set_action(self, action, agent, action_space, time=None):
        """environment.py:144-192 with every scalar widened to a [B, .] tensor."""
        w, B, dev = self.world, self.batch_size, self.world.device
        agent.action.u = torch.zeros((B, w.dim_p), dtype=torch.float32, device=dev)
        agent.action.c = torch.zeros((B, w.dim_c), dtype=torch.float32, device=dev)
        if not torch.is_tensor(action):
            action = torch.as_tensor(np.asarray(action), device=dev)
        action = action.to(dev)
        if isinstance(action_space, spaces.MultiDiscrete):
            if self.discrete_action_input:
                act = [action.reshape(B, -1)[:, k] for k in range(action_space.num_discrete_space)]
            else:
                sizes = [int(s) for s in (action_space.high - action_space.low + 1)]
                flat = action.reshape(-1, sum(sizes)).expand(B, sum(sizes))
                act, index = [], 0
                for s in sizes:
                    act.append(flat[:, index:index + s])
                    index += s
        else:
            act = [action]
        if agent.movable:
            if self.discrete_action_input:
                k = act[0].reshape(-1).expand(B)
                u = agent.action.u
                u[:, 0] = (k == 2).float() - (k == 1).float()   # 1:-x 2:+x (environment.py:164-165)
                u[:, 1] = (k == 4).float() - (k == 3).float()   # 3:-y 4:+y
            else:
                a = act[0].float().reshape(-1, act[0].shape[-1]).expand(B, act[0].shape[-1])
                if self.force_discrete_action:
                    d = a.argmax(dim=-1, keepdim=True)
                    a = torch.zeros_like(a).scatter_(-1, d, 1.0)
                if self.discrete_action_space:
                    agent.action.u[:, 0] += a[:, 1] - a[:, 2]
                    agent.action.u[:, 1] += a[:, 3] - a[:, 4]
                else:
                    agent.action.u = a.clone()
            sensitivity = 5.0
            if agent.accel is not None:
                sensitivity = agent.accel
            agent.action.u *= sensitivity
            act = act[1:]
        if not agent.silent:
            if self.discrete_action_input:
                k = act[0].reshape(-1).expand(B).long()
                agent.action.c = torch.zeros((B, w.dim_c), dtype=torch.float32, device=dev)
                agent.action.c.scatter_(1, k[:, None], 1.0)
            else:
                agent.action.c = act[0].float().reshape(-1, w.dim_c).expand(B, w.dim_c).clone()
            act = act[1:]
        assert len(act) == 0

    def _step_generic(self, action_n):
        if isinstance(action_n, tuple) and len(action_n) == 2 and all(torch.is_tensor(t) and t.dim() == 3 for t in action_n):
            raise _abi.MpeError("the (moves, utterances) tuple form is decoded by the fused step; this env steps through Python "
                                "callbacks / another action mode: pass per-agent [move | utterance] rows")
        for i, agent in enumerate(self.agents):
            self._set_action(action_n[i], agent, self.action_space[i])
        self.world.step()
        if self._prog is not None and self.fused:
            # an action mode the physics launch does not decode (integer ids with communication, continuous forces): the
            # actions were decoded in torch and World.step ran on its own; rows and rewards are still the program's
            self._ensure_buffers()
            for i, agent in enumerate(self.world.agents):
                if self._comm is not None and not agent.silent and torch.is_tensor(agent.state.c):
                    self._comm[i].copy_(agent.state.c)
            out = self._next_set()
            b = out.bufs
            b.act = b.ids = b.u = None
            _abi.check(_abi.lib().mpe_rows(C.byref(self._desc), out.bufs_ref, self._prog.ref, self.batch_size, self._stream()), "mpe_rows")
            done_n, done = out.done_n, out.done
            if self._py_done:
                done = torch.stack([torch.as_tensor(self._get_done(a), device=self.world.device).bool().expand(self.batch_size)
                                    for a in self.agents]).contiguous()
                done_n = [done[i] for i in range(self.n)]
            info_n = {'n': [self._get_info(a) for a in self.agents]}
            if self.max_episode_steps and self._episode_tick(done, out):
                self._observe_into(out)
            return self._deliver(out.obs_n, out.reward_n, done_n, info_n)
        obs_n, reward_n, done_n, info_n = [], [], [], {'n': []}
        for agent in self.agents:
            obs_n.append(self._get_obs(agent))
            reward_n.append(self._get_reward(agent))
            done_n.append(self._get_done(agent))
            info_n['n'].append(self._get_info(agent))
        if self.shared_reward:  # environment.py:100-102: every agent gets the sum
            total = torch.stack([torch.as_tensor(r, device=self.world.device) for r in reward_n]).sum(dim=0)
            reward_n = [total] * self.n
        if self.max_episode_steps:
            done = torch.stack([torch.as_tensor(d, device=self.world.device).bool().expand(self.batch_size)
                                for d in done_n]).contiguous()
            if self._episode_tick(done):
                obs_n = [self._get_obs(agent) for agent in self.agents]
            done_n = [done[i] for i in range(len(done_n))]
        return self._deliver(obs_n, reward_n, done_n, info_n)

    # rendering is a GUI concern of the reference (environment.py:200-263) and is not provided
    def render(self, mode='human'):
        raise NotImplementedError("rendering is out of scope of the MI355X hot-path build (see DESIGN.md)")


class GraphedStep(object):
    """`env.step` captured once into a HIP graph and replayed: the host leaves the per-step loop.

    The generic path (a user Scenario's torch callbacks around `mpe_world_step`) is a hundred small launches per
    step, each costing the Python host microseconds; replaying them as one graph costs one launch call.  Nothing is
    traced or compiled: the same kernels run in the same order on the same buffers.

        g = GraphedStep(env, example_action_n)      # captures env.step(example)
        obs_n, rew_n, done_n, info_n = g.step(action_n)       # copies action_n into g.action_n, replays
        g.action_n[i].copy_(policy_output_i); g.step()        # or write the static inputs yourself

    The returned tensors are the graph's static outputs (overwritten by the next replay).  Callbacks must be
    capturable: no .item()/.cpu()/host branches on device data.  Episode bookkeeping (max_episode_steps) decides
    on the host when to launch resets, so it is not captured: reset between replays with env.reset(...)."""

    def __init__(self, env, action_n, warmup=3):
        if env.numpy_io:
            raise _abi.MpeError("GraphedStep works on device tensors (make_env(..., batch_size=B))")
        if env.max_episode_steps:
            raise _abi.MpeError("GraphedStep does not capture episode bookkeeping; count steps outside or use auto-reset eagerly")
        self.env = env
        dev = env.world.device
        as_list = not torch.is_tensor(action_n)
        self.action_n = [torch.as_tensor(a, device=dev).clone() for a in action_n] if as_list else action_n.clone()
        fresh, env.fresh_outputs = env.fresh_outputs, True     # the graph owns its outputs
        pos, vel = env.world.pos.clone(), env.world._vel_all.clone()     # every entity's row: movable landmarks are integrated too
        comm = [a.state.c.clone() if torch.is_tensor(a.state.c) else a.state.c for a in env.world.agents]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):            # allocator / code-object warm-up outside the capture
                    env.step(self.action_n)
                torch.cuda.synchronize(dev)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=side):
                    self.outputs = env.step(self.action_n)
        finally:
            env.fresh_outputs = fresh
        torch.cuda.current_stream(dev).wait_stream(side)
        self._comm_state = [a.state.c for a in env.world.agents]   # the tensors the graph writes (resets re-point state.c)
        env.world.pos.copy_(pos)                   # the warm-up steps did not happen
        env.world._vel_all.copy_(vel)
        for a, c in zip(env.world.agents, comm):
            if torch.is_tensor(c) and torch.is_tensor(a.state.c):
                a.state.c.copy_(c)

    def step(self, action_n=None):
        if action_n is not None:
            if torch.is_tensor(self.action_n):
                self.action_n.copy_(action_n)
            else:
                for dst, src in zip(self.action_n, action_n):
                    dst.copy_(torch.as_tensor(src, device=dst.device))
        self.graph.replay()
        for a, c in zip(self.env.world.agents, self._comm_state):
            a.state.c = c
        return self.outputs


class BatchMultiAgentEnv(object):
    """The reference's list-of-envs wrapper (environment.py:288-335): per-agent lists of several envs
    concatenated, `n` = total number of agents.  Each member may itself be a batched env (the batch
    axis is the efficient way to run many worlds; this wrapper only keeps the reference's calling
    convention for code written against it).  The reference's `step` forwards an extra `time` argument
    that `MultiAgentEnv.step` does not take (it never worked as shipped); here `time` is
    accepted and ignored."""
    metadata = {'runtime.vectorized': True, 'render.modes': []}

    def __init__(self, env_batch):
        self.env_batch = list(env_batch)

    @property
    def n(self):
        return int(sum(env.n for env in self.env_batch))

    @property
    def action_space(self):
        return self.env_batch[0].action_space

    @property
    def observation_space(self):
        return self.env_batch[0].observation_space

    def step(self, action_n, time=None):
        obs_n, reward_n, done_n, info_n = [], [], [], {'n': []}
        i = 0
        for env in self.env_batch:
            obs, reward, done, _ = env.step(action_n[i:(i + env.n)])
            i += env.n
            obs_n += obs
            reward_n += reward
            done_n += done
        return obs_n, reward_n, done_n, info_n

    def reset(self):
        obs_n = []
        for env in self.env_batch:
            obs_n += env.reset()
        return obs_n

    def render(self, mode='human', close=True):
        raise NotImplementedError("rendering is out of scope of the MI355X hot-path build (see DESIGN.md)")
