"""Batched world model: the reference's Entity/Agent/Landmark/World API over SoA device tensors.

The reference keeps one Python object per entity, each holding its own 2-vector
(multiagent/core.py:4-79).  Here a World owns the state of B independent worlds as two
structure-of-arrays tensors with the batch index innermost,

    world.pos  [E, 2, B] fp32     agents first, then landmarks   (core.py:103-104 ordering)
    world.vel  [A, 2, B] fp32     the agents' rows of an [E, 2, B] velocity block: in the shipped scenarios landmarks are
                                  never integrated (core.py:160, movable = False); a landmark a user makes movable is
                                  integrated like any entity (core.py:158-169) and its velocity lives in the rows behind

which is the layout the HIP kernels read with coalesced 256-byte wave accesses.  The familiar
attribute paths still work: `agent.state.p_pos` is a live [B, 2] *view* of that storage and
assigning to it copies into the storage, so scenario code written against the reference's names
(`entity.state.p_pos - agent.state.p_pos`) runs unchanged on batched tensors.

World.step() = core.py:117-131 executed by libmpe_hip.so (`mpe_world_step`), never on the CPU.
"""
import ctypes as C

import torch

from . import _abi


class EntityState(object):
    """p_pos / p_vel of one entity for all B worlds (reference: core.py:4-9)."""

    def __init__(self):
        self._world = None
        self._index = None
        self._is_agent = False
        self._pending = {}

    def _bind(self, world, index, is_agent):
        self._world, self._index, self._is_agent = world, index, is_agent
        for k, v in self._pending.items():
            setattr(self, k, v)
        self._pending = {}

    @property
    def p_pos(self):
        if self._world is None:
            return self._pending.get("p_pos")
        return self._world.pos[self._index].t()

    @p_pos.setter
    def p_pos(self, value):
        if self._world is None:
            self._pending["p_pos"] = value
            return
        self._world.pos[self._index].t().copy_(self._world._as_batch(value, 2))

    @property
    def p_vel(self):
        if self._world is None:
            return self._pending.get("p_vel")
        return self._world._vel_all[self._index].t()   # (a landmark's row stays zero unless it is movable)

    @p_vel.setter
    def p_vel(self, value):
        if self._world is None:
            self._pending["p_vel"] = value
            return
        if not self._is_agent and not self._world.entities[self._index].movable:
            return  # an immovable landmark's velocity is identically zero (never integrated, never observed)
        self._world._vel_all[self._index].t().copy_(self._world._as_batch(value, 2))


class AgentState(EntityState):
    """Adds the communication utterance c (reference: core.py:12-16)."""

    def __init__(self):
        super(AgentState, self).__init__()
        self.c = None


class Action(object):
    """Physical action u [B,2] and communication action c (reference: core.py:19-24)."""

    def __init__(self):
        self.u = None
        self.c = None


# attributes the kernels' descriptor snapshots (MpeScenarioDesc): assigning any of them on a bound entity / world
# marks the snapshot stale, and the next step re-reads them -- the reference reads them live on every step
_ENTITY_CONSTANTS = frozenset(["size", "movable", "collide", "max_speed", "accel", "initial_mass", "u_noise", "c_noise",
                               "silent"])
_WORLD_CONSTANTS = frozenset(["dt", "damping", "contact_force", "contact_margin", "collaborative", "dim_c"])


class Entity(object):
    """Per-entity constants, identical across the B worlds (reference: core.py:27-51)."""

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name in _ENTITY_CONSTANTS:
            w = getattr(self.__dict__.get("state"), "_world", None)
            if w is not None:
                w._constants_version += 1

    def __init__(self):
        self.name = ''
        self.size = 0.050
        self.movable = False
        self.collide = True
        self.density = 25.0
        self.color = None
        self.max_speed = None
        self.accel = None
        self.state = EntityState()
        self.initial_mass = 1.0

    @property
    def mass(self):
        return self.initial_mass


class Landmark(Entity):
    def __init__(self):
        super(Landmark, self).__init__()


class Agent(Entity):
    """reference: core.py:59-79."""

    def __init__(self):
        super(Agent, self).__init__()
        self.movable = True
        self.silent = False
        self.blind = False
        self.u_noise = None
        self.c_noise = None
        self.u_range = 1.0
        self.state = AgentState()
        self.action = Action()
        self.action_callback = None


class EntityChoice(object):
    """A per-world pick among `entities` -- what the reference writes as
    `agent.goal_a = np.random.choice(world.landmarks)` (simple_adversary.py:44, simple_push.py:41, ...).
    `index` is a [B] integer tensor -- pass a VIEW of persistent storage (`world.choice_i32[k]`) so that resets, which
    rewrite it in place, are seen by a captured graph; attribute access gathers from the picked entity of every world, so
    reference-style scenario code (`agent.goal_a.state.p_pos`, `agent.goal_b.color`) runs unchanged."""

    def __init__(self, world, entities, index):
        self._world = world
        self.entities = list(entities)
        self.index = index

    def _gather(self, per_entity):
        stack = torch.stack(per_entity, dim=0)                      # [K, B, w]
        ar = torch.arange(stack.shape[1], device=stack.device)
        return stack[self.index.to(stack.device).long(), ar]

    @property
    def state(self):
        return self

    @property
    def p_pos(self):
        return self._gather([e.state.p_pos for e in self.entities])

    @property
    def p_vel(self):
        return self._gather([e.state.p_vel for e in self.entities])

    @property
    def color(self):
        return self._gather([self._world._as_batch(e.color, len(e.color) if not torch.is_tensor(e.color)
                                                   else e.color.shape[-1]) for e in self.entities])

    @property
    def size(self):
        sizes = [e.size for e in self.entities]
        if all(x == sizes[0] for x in sizes):
            return sizes[0]
        return self._world.constant(sizes)[self.index.long()]


def _i64(x):
    """A 64-bit pattern as the signed value torch.int64 holds."""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def _mix64(z):
    """SplitMix64's finaliser on int64 tensors (wrapping multiplies, logical shifts spelled out)."""
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _i64(0xBF58476D1CE4E5B9)
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _i64(0x94D049BB133111EB)
    return z ^ ((z >> 31) & ((1 << 33) - 1))


def counter_randn(seed, world_offset, draw, shape, device):
    """[B, k] standard normals as a pure function of (seed, global world index, draw number, column): Box-Muller over
    two 24-bit uniforms hashed from those four numbers.  Shard-invariant by construction."""
    B, k = int(shape[0]), int(shape[1]) if len(shape) > 1 else 1
    w = torch.arange(B, dtype=torch.int64, device=device).unsqueeze(1) + int(world_offset)
    j = torch.arange(k, dtype=torch.int64, device=device).unsqueeze(0)
    x = w * _i64(0x9E3779B97F4A7C15) + j * _i64(0xD1B54A32D192ED03) + _i64(seed * 0xA0761D6478BD642F + draw * 0xE7037ED1A0B428DB)
    h1 = _mix64(x + _i64(0x9E3779B97F4A7C15))
    h2 = _mix64(h1 + _i64(0x9E3779B97F4A7C15))
    u1 = (((h1 >> 40) & 0xFFFFFF).to(torch.float32) + 1.0) * (1.0 / 16777216.0)     # (0, 1]
    u2 = ((h2 >> 40) & 0xFFFFFF).to(torch.float32) * (1.0 / 16777216.0)             # [0, 1)
    n = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(6.283185307179586 * u2)
    return n.reshape(shape)


class World(object):
    """B particle worlds stepped in lock-step (reference: core.py:82-196 for one world)."""

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name in _WORLD_CONSTANTS and "_constants_version" in self.__dict__:
            self.__dict__["_constants_version"] += 1
            self.__dict__["_desc"] = None
            self.__dict__["_entity_table"] = None

    def __init__(self, batch_size=1, device=None):
        self._constants_version = 0   # bumped by every assignment to a constant the kernel descriptor snapshots
        self._desc_version = -1
        self.agents = []
        self.landmarks = []
        self.dim_c = 0
        self.dim_p = 2
        self.dim_color = 3
        self.dt = 0.1
        self.damping = 0.25
        self.contact_force = 1e+2
        self.contact_margin = 1e-3
        self.batch_size = int(batch_size)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
                else torch.device("cpu")
        self.device = torch.device(device)
        self.pos = None
        self.vel = None
        self._zero_vel = None
        self._desc = None
        self._entity_table = None
        self._u = None
        self._bufs = None
        self.rng_mode = "device"   # 'device' (Philox, mpe_reset) | 'numpy' (reference-order global np.random)
        self.seed = 0
        self.world_offset = 0      # global index of this batch's world 0 (multi-GPU sharding by batch index)
        self._episode = 0
        self.choice_pops = []      # population sizes of the per-world picks reset_world draws (goal landmark, ...)
        self.choice_i32 = None     # [K, B] int32 device tensor holding them (what the fused kernels read)

    # ---- reference properties (core.py:101-114) --------------------------------------------------
    @property
    def entities(self):
        return self.agents + self.landmarks

    @property
    def policy_agents(self):
        return [agent for agent in self.agents if agent.action_callback is None]

    @property
    def scripted_agents(self):
        return [agent for agent in self.agents if agent.action_callback is not None]

    # ---- SoA storage -----------------------------------------------------------------------------
    def allocate(self):
        """Create the SoA tensors and bind every entity's state to its slice.  Called by
        Scenario.make_world once the agent/landmark lists are final ("no agents created or
        destroyed at runtime", environment.py:8)."""
        A, E, B = len(self.agents), len(self.entities), self.batch_size
        self.pos = torch.zeros((E, 2, B), dtype=torch.float32, device=self.device)
        self._vel_all = torch.zeros((E, 2, B), dtype=torch.float32, device=self.device)
        self.vel = self._vel_all[:A]       # what the fused kernels read: [A][2][B], the head of the block
        self._zero_vel = torch.zeros((B, 2), dtype=torch.float32, device=self.device)
        for i, ent in enumerate(self.entities):
            ent.state._bind(self, i, i < A)
        for agent in self.agents:   # agent.state.c = np.zeros(world.dim_c), as every reset_world leaves it
            agent.state.c = torch.zeros((B, self.dim_c), dtype=torch.float32, device=self.device)
        if self.choice_pops:
            self.choice_i32 = torch.zeros((len(self.choice_pops), B), dtype=torch.int32, device=self.device)
        self._desc = None
        return self

    def constant(self, values):
        """A small host constant (a colour, a list of sizes) as a cached fp32 device tensor: uploaded once, so
        callbacks neither pay a host-to-device copy per step nor break HIP-graph capture (GraphedStep)."""
        import numpy as np
        arr = np.asarray(values, dtype=np.float32)
        key = (arr.shape, arr.tobytes())
        cache = self.__dict__.setdefault("_constants", {})
        t = cache.get(key)
        if t is None or t.device != torch.device(self.device):
            t = cache[key] = torch.as_tensor(arr).to(self.device)
        return t

    def _as_batch(self, value, width):
        t = value.to(dtype=torch.float32, device=self.device) if torch.is_tensor(value) else self.constant(value)
        if t.dim() == 1:
            t = t.unsqueeze(0)
        return t.expand(self.batch_size, width)

    def set_state(self, pos, vel=None):
        """Upload states: pos [B, E, 2], vel [B, A, 2] -- or [B, E, 2] when landmarks move -- (host order, as the
        oracle keeps them)."""
        p = torch.as_tensor(pos, dtype=torch.float32).reshape(self.batch_size, len(self.entities), 2)
        self.pos.copy_(p.permute(1, 2, 0))
        self._vel_all.zero_()
        if vel is not None:
            v = torch.as_tensor(vel, dtype=torch.float32)
            v = v.reshape(self.batch_size, v.numel() // (2 * self.batch_size), 2)
            if v.shape[1] not in (len(self.agents), len(self.entities)):
                raise _abi.MpeError("set_state: vel holds %d entities (agents: %d, all: %d)" % (v.shape[1], len(self.agents), len(self.entities)))
            self._vel_all[:v.shape[1]].copy_(v.permute(1, 2, 0))

    def get_state(self, all_entities=None):
        """(pos [B,E,2], vel [B,A,2]) as host NumPy arrays; all_entities: vel [B,E,2].  Default: the agents' rows -- and
        every entity's as soon as a landmark is movable (its velocity is state, core.py:158-169), so that
        `set_state(*get_state())` restores the world either way."""
        if all_entities is None:
            all_entities = self.n_dynamic > len(self.agents)
        v = self._vel_all if all_entities else self.vel
        return (self.pos.permute(2, 0, 1).contiguous().cpu().numpy(), v.permute(2, 0, 1).contiguous().cpu().numpy())

    @property
    def n_dynamic(self):
        """Entities [0, n_dynamic) are what World.step integrates: the agents, and the landmarks up to the last MOVABLE
        one (core.py:158-169 integrates every movable entity; `mpe_world_step` takes those as action-less agents)."""
        n = len(self.agents)
        for e, ent in enumerate(self.entities):
            if e >= n and ent.movable:
                n = e + 1
        return n

    # ---- reset_world bodies shared by the built-in scenarios ---------------------------------------
    def reset_boxes(self, boxes, mask=None, choices=None, seeds=None):
        """reset_uniform with a placement box per entity: `boxes[e] = (lo_x, hi_x, lo_y, hi_y)` -- entity e's position is
        `np.random.uniform(lo, hi)` per coordinate (a restricted spawn area, landmarks off-centre).  On the device the draws are
        `mpe_reset_rows`' -- the ones a row program with the same `reset_boxes` (rowspec.RowProgram, Scenario.reset_boxes) makes
        for its in-launch restarts and rollouts."""
        if len(boxes) != len(self.entities):
            raise _abi.MpeError("reset_boxes: one (lo_x, hi_x, lo_y, hi_y) per entity")
        if len(boxes) > _abi.MPE_ROWS_MAX_ENTITIES:
            raise _abi.MpeError("reset_boxes: %d entities (per-entity boxes cover %d)" % (len(boxes), _abi.MPE_ROWS_MAX_ENTITIES))
        return self.reset_uniform(1.0, mask, choices, seeds, boxes=[tuple(float(v) for v in b) for b in boxes])

    def reset_uniform(self, landmark_range=1.0, mask=None, choices=None, seeds=None, redraw=None, boxes=None):
        """Scenario.reset_world for the shipped scenarios (simple_spread.py:38-45, simple_tag.py:46-54,
        simple.py:33-39, ...): agents ~ U[-1,1)^2, landmarks ~ U[-r,r)^2, vel = 0, comm state c = 0.
        `choices` = [n_0, n_1, ...]: the `np.random.choice` draws the scenario makes BEFORE the positions
        (goal landmarks, simple_adversary.py:44 ...); returned as a [B, len(choices)] long tensor.
        `redraw` = entity indices the scenario draws a SECOND time after the main pass
        (simple_world_comm.py:108-113 places food and forests twice): it only matters for reproducing
        the reference's random stream (host modes); the distribution is the same.
        Where the random numbers come from:
          seeds given       world b is drawn from np.random.RandomState(seeds[b]) in the reference's order
                            -- exactly `np.random.seed(s); env.reset()` per world (the parity-test reset);
          rng_mode 'numpy'  the process-global np.random in the reference's order, world by world
                            (seed-identical to the reference for B = 1: compatibility mode);
          rng_mode 'device' Philox on the GPU (`mpe_reset`), keyed by (seed, world, episode); choices from
                            a torch generator keyed the same way."""
        import numpy as np
        A, E, B = len(self.agents), len(self.entities), self.batch_size
        choices = list(choices or [])
        idx = None
        if seeds is not None or self.rng_mode == "numpy":
            m = None if mask is None else torch.as_tensor(mask).cpu().numpy().astype(bool)
            if seeds is not None:
                assert len(seeds) == B
            pos, vel = self.get_state(all_entities=True) if m is not None else (np.zeros((B, E, 2), np.float64), np.zeros((B, E, 2)))
            pos = pos.astype(np.float64)
            idx_np = np.zeros((B, len(choices)), np.int64)
            for b in range(B):
                if m is not None and not m[b]:
                    continue
                rs = np.random.RandomState(int(seeds[b])) if seeds is not None else np.random
                for k, n in enumerate(choices):
                    idx_np[b, k] = rs.randint(0, n)        # == np.random.choice(list of n) (same stream)
                for e in list(range(E)) + list(redraw or []):
                    if boxes is not None:
                        pos[b, e] = (rs.uniform(boxes[e][0], boxes[e][1]), rs.uniform(boxes[e][2], boxes[e][3]))
                        continue
                    r = 1.0 if e < A else landmark_range
                    pos[b, e] = rs.uniform(-r, +r, self.dim_p)
                vel[b] = 0.0
            self.set_state(pos, vel)
            idx = torch.as_tensor(idx_np, device=self.device)
        else:
            self._require_device()
            desc = self.scenario_desc(_abi.MPE_SCN_GENERIC)
            bufs = _abi.MpeBuffers()
            bufs.pos, bufs.vel = self.pos.data_ptr(), self.vel.data_ptr()
            drawn = None
            if choices:   # drawn by the same kernel, keyed like the positions (and like the fused rollout's in-kernel resets)
                assert choices == list(self.choice_pops), "declare world.choice_pops before allocate()"
                drawn = self.choice_i32.clone()
                bufs.choice = drawn.data_ptr()
            mptr = None
            if mask is not None:
                mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
                mptr = C.c_void_p(mask.data_ptr())
            if boxes is not None:
                p = _abi.MpeRowProgram()
                p.reset_boxes = 1
                for e, (lx, hx, ly, hy) in enumerate(boxes):
                    p.reset_box[e][0], p.reset_box[e][1], p.reset_box[e][2], p.reset_box[e][3] = lx, hx - lx, ly, hy - ly
                _abi.check(_abi.lib().mpe_reset_rows(C.byref(desc), C.byref(bufs), C.byref(p), B, mptr, 1.0,
                                                     int(self.seed) & (2 ** 64 - 1), int(self._episode), int(self.world_offset),
                                                     _abi.raw_stream(self.device)), "mpe_reset_rows")
            else:
                _abi.check(_abi.lib().mpe_reset(C.byref(desc), C.byref(bufs), B, mptr, float(landmark_range),
                                                int(self.seed) & (2 ** 64 - 1), int(self._episode), int(self.world_offset),
                                                _abi.raw_stream(self.device)),
                           "mpe_reset")
            if choices:
                idx = drawn.t().long()
            self._episode += 1
        # comm state of every agent starts at zero (agent.state.c = np.zeros(world.dim_c) in every reset_world)
        keep = None if mask is None else ~torch.as_tensor(mask, device=self.device).bool()
        if self.n_dynamic > A:     # landmark.state.p_vel = np.zeros(world.dim_p) (every reset_world): the movable ones' rows
            if keep is None:
                self._vel_all[A:].zero_()
            else:
                self._vel_all[A:].mul_(keep.to(torch.float32)[None, None, :])
        for agent in self.agents:
            z = torch.zeros((B, self.dim_c), dtype=torch.float32, device=self.device)
            if keep is not None and torch.is_tensor(agent.state.c):
                z = torch.where(keep[:, None], agent.state.c, z)
            agent.state.c = z
        return idx if choices else None

    def reset_from_numpy_seeds(self, seeds, landmark_range=1.0):
        """World b starts exactly as the reference would after `np.random.seed(seeds[b]);
        env.reset()` (MT19937 draws on the host, then one upload) -- the parity-test reset."""
        return self.reset_uniform(landmark_range, seeds=seeds)

    @staticmethod
    def merge_choice(old, new, mask):
        """Per-world choice indices after a partial reset: new where mask, old elsewhere."""
        if mask is None or old is None:
            return new
        return torch.where(torch.as_tensor(mask, device=new.device).bool(), new, old)

    # ---- descriptor handed to the C ABI ----------------------------------------------------------
    def scenario_desc(self, kind=_abi.MPE_SCN_GENERIC, n_adversaries=0):
        """MpeScenarioDesc for this world (cached per kind).  Constants are read off the entity
        objects exactly where the reference reads them (size/movable/collide/accel/max_speed/mass,
        core.py:27-51; accel -> sensitivity default 5.0, environment.py:178-181)."""
        key = (kind, n_adversaries)
        if self._desc_version != self._constants_version:   # an entity / world constant was assigned since the snapshot
            self._desc = None
            self._entity_table = None
            self._desc_version = self._constants_version
        if self._desc is None:
            self._desc = {}
        if key in self._desc:
            return self._desc[key]
        ents = self.entities
        A, L = len(self.agents), len(self.landmarks)
        if A + L > _abi.MPE_MAX_ENTITIES:
            raise _abi.MpeError("too many entities: %d > %d" % (A + L, _abi.MPE_MAX_ENTITIES))
        d = _abi.MpeScenarioDesc()
        d.kind, d.n_agents, d.n_landmarks, d.dim_c = kind, A, L, int(self.dim_c)
        d.n_adversaries = int(n_adversaries)
        d.collaborative = 1 if getattr(self, "collaborative", False) else 0
        d.dt, d.damping = self.dt, self.damping
        d.contact_force, d.contact_margin = self.contact_force, self.contact_margin
        for e, ent in enumerate(ents):
            d.size[e] = ent.size
            d.mass[e] = ent.mass
            d.accel[e] = 5.0 if ent.accel is None else ent.accel
            d.max_speed[e] = -1.0 if ent.max_speed is None else ent.max_speed
            d.movable[e] = 1 if ent.movable else 0
            d.collide[e] = 1 if ent.collide else 0
        d.n_choices = len(self.choice_pops)
        for k, n in enumerate(self.choice_pops):
            d.choice_pop[k] = int(n)
        if kind != _abi.MPE_SCN_GENERIC:
            total = _abi.lib().mpe_fill_obs_layout(C.byref(d))
            if total < 0:
                _abi.check(total, "mpe_fill_obs_layout")
        self._desc[key] = d
        return d

    def entity_table(self, desc):
        """Device copy of the per-entity constant table (the wave-per-world kernel for large N reads it)."""
        if self._entity_table is None:
            n = _abi.lib().mpe_fill_entity_table(C.byref(desc), None)
            host = (C.c_float * n)()
            _abi.lib().mpe_fill_entity_table(C.byref(desc), host)
            self._entity_table = torch.tensor(list(host), dtype=torch.float32, device=self.device)
        return self._entity_table

    def _require_device(self):
        if self.pos is None:
            raise _abi.MpeError("World.allocate() has not been called")
        if not self.pos.is_cuda:
            raise _abi.MpeError("the step path runs on a HIP device only (world tensors are on %s); "
                                "there is no CPU fallback" % self.pos.device)

    def _randn(self, shape):
        """Standard-normal noise for u_noise / c_noise (core.py:138,176).  rng_mode 'numpy' (reference-compatibility):
        drawn from the process-global np.random in the reference's order (agents in order: the u draws before the
        physics, the c draws after), so `np.random.seed(s)` reproduces the reference's noisy trajectories and leaves
        the stream where the reference leaves it for the next reset.  rng_mode 'device': counter-based like the resets
        and the synthetic moves -- the value for (seed, GLOBAL world, draw number, element) is a hash of exactly those,
        so a batch sharded over several GPUs (world_offset) draws what one big batch would, whatever the shard sizes;
        independent of torch's global generator."""
        if self.rng_mode == "numpy":
            import numpy as np
            return torch.as_tensor(np.random.randn(*shape), dtype=torch.float32).to(self.device)
        draw = self.__dict__.get("_noise_draw", 0)
        self.__dict__["_noise_draw"] = draw + 1
        return counter_randn(int(self.seed), int(self.world_offset), draw, tuple(shape), self.device)

    # ---- World.step (core.py:117-131) -----------------------------------------------------------
    def step(self):
        """Advance all B worlds using each agent's `action.u` ([B,2] tensors), physics only:
        apply_action_force, apply_environment_force, integrate_state, update_agent_state."""
        self._require_device()
        for agent in self.scripted_agents:
            agent.action = agent.action_callback(agent, self)
        A, B = len(self.agents), self.batch_size
        nd = self.n_dynamic
        if self._u is None or self._u.shape[0] != nd:
            self._u = torch.zeros((nd, 2, B), dtype=torch.float32, device=self.device)   # rows [A, nd): movable landmarks, no action
        for i, agent in enumerate(self.agents):
            if agent.movable and agent.action.u is not None:
                u = self._as_batch(agent.action.u, 2)
                if agent.u_noise:   # core.py:138: np.random.randn(*agent.action.u.shape) * agent.u_noise
                    u = u + self._randn(u.shape) * agent.u_noise
                self._u[i].t().copy_(u)
            else:
                self._u[i].zero_()
        desc = self.scenario_desc(_abi.MPE_SCN_GENERIC)
        bufs = _abi.MpeBuffers()
        bufs.pos, bufs.vel, bufs.u = self.pos.data_ptr(), self._vel_all.data_ptr(), self._u.data_ptr()
        bufs.entity_table = self.entity_table(desc).data_ptr()
        _abi.check(_abi.lib().mpe_world_step(C.byref(desc), C.byref(bufs), B, _abi.raw_stream(self.device)), "mpe_world_step")
        for agent in self.agents:  # update_agent_state (core.py:171-177)
            if agent.silent:
                agent.state.c = torch.zeros((B, self.dim_c), dtype=torch.float32, device=self.device)
            else:
                c = agent.action.c
                if c is not None and agent.c_noise:   # core.py:176
                    c = c + self._randn(c.shape) * agent.c_noise
                agent.state.c = c
