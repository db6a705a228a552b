"""ObsSpec / RewardSpec: a scenario's observation and reward as DATA the `mpe_rows` kernel interprets.

The reference's plug-in promise is that writing a new scenario is the normal use (README "Creating new environments",
scenario.py:4-10).  With the torch protocol a new scenario's callbacks are Python: a hundred-odd small launches per
step.  But every observation the reference ships is a concatenation of a few SEGMENT kinds, and every reward an
ordered sum of a few TERM kinds (simple_spread.py:72-100, simple_tag.py:84-147, simple_world_comm.py:143-289 ...), so a
scenario can say WHAT its rows are instead of computing them:

    class Scenario(BaseScenario):
        def make_world(self, batch_size=1, device=None): ...
        def reset_world(self, world, mask=None, seeds=None): ...
        def obs_spec(self, agent, world):                 # instead of (or next to) observation(agent, world)
            o = ObsSpec(world, agent)
            o.vel().pos()
            for lm in world.landmarks: o.rel(lm)
            for other in world.agents:
                if other is not agent: o.rel(other)
            return o
        def reward_spec(self, agent, world):              # instead of (or next to) reward(agent, world)
            r = RewardSpec(world, agent)
            for lm in world.landmarks:
                r.min_dist([a for a in world.agents], lm).add(-1.0)
            for a in world.agents:
                r.add_if_touching(a, agent, -1.0)
            return r

`MultiAgentEnv` then steps the scenario in TWO launches -- `mpe_world_step` (World.step) + `mpe_rows` (every agent's
row, reward, done) -- whatever the scenario is; no JIT, no generated code: the specs compile to a list of 16-byte ops
(include/mpe_hip.h, enum MpeRowOp) uploaded once.  Arithmetic happens in program order with the kernels' own device
functions, so a spec that follows a reference callback's order reproduces it to the fp32 bit; `builtin_specs` below
holds the nine shipped scenarios written this way (tests/test_rowspec.py: bit-identical to their fused kernels).

Entities are named by the objects themselves (`world.agents[k]`, `world.landmarks[k]`) or by index into
`world.entities`; per-world picks (`agent.goal_a = np.random.choice(world.landmarks)`) by their row of
`world.choice_i32`.
"""
import ctypes as C
import struct

import torch

from . import _abi

SELF = _abi.MPE_ROW_SELF
FUSE = True      # RowProgram's default: fuse runs of per-entity ops into range forms (tests switch it off for the A/B)


def _f2i(x):
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def _code_op(code, w1, source_hash):
    h = int(source_hash) & (2 ** 64 - 1)
    lo, hi = h & 0xFFFFFFFF, h >> 32
    return (int(code), int(w1), lo - (1 << 32) if lo >= 1 << 31 else lo, hi - (1 << 32) if hi >= 1 << 31 else hi)


def _op(code, a0=0, a1=0, a2=0, w1=0, f0=0.0, f1=0.0):
    for v in (code, a0, a1, a2):
        if not 0 <= int(v) <= 255:
            raise _abi.MpeError("row program: argument %r does not fit a byte" % (v,))
    w0 = int(code) | (int(a0) << 8) | (int(a1) << 16) | (int(a2) << 24)
    if w0 >= 1 << 31:
        w0 -= 1 << 32
    return (w0, int(w1), _f2i(f0), _f2i(f1))


class _Spec(object):
    def __init__(self, world, agent=None):
        self.world = world
        self.agent = agent
        self.ops = []
        self._ents = world.entities

    def _e(self, ent, self_ok=False):
        """An entity object / index -> its index in world.entities."""
        if ent is None or (self_ok and ent is self.agent):
            if self.agent is None:
                raise _abi.MpeError("spec: no observing agent to default to")
            return SELF if self_ok else self._ents.index(self.agent)
        if isinstance(ent, int):
            if not 0 <= ent < len(self._ents):
                raise _abi.MpeError("spec: entity index %d out of range" % ent)
            return ent
        for k, e in enumerate(self._ents):
            if e is ent:
                return k
        raise _abi.MpeError("spec: %r is not an entity of this world" % (ent,))

    def _a(self, agent):
        k = self._e(agent)
        if k >= len(self.world.agents):
            raise _abi.MpeError("spec: entity %d is not an agent" % k)
        return k


class ObsSpec(_Spec):
    """One agent's observation row as a list of segments, in the order they are concatenated."""

    def __init__(self, world, agent):
        super(ObsSpec, self).__init__(world, agent)
        self.width = 0

    def _emit(self, op, w):
        self.ops.append(op)
        self.width += w
        return self

    def vel(self, ent=None):
        """entity.state.p_vel (default: the observing agent's) -- 2 columns."""
        return self._emit(_op(_abi.MPE_ROW_OBS_VEL, self._e(ent, True)), 2)

    def pos(self, ent=None):
        """entity.state.p_pos -- 2 columns."""
        return self._emit(_op(_abi.MPE_ROW_OBS_POS, self._e(ent, True)), 2)

    def rel(self, ent):
        """ent.state.p_pos - agent.state.p_pos -- 2 columns."""
        return self._emit(_op(_abi.MPE_ROW_OBS_REL, self._e(ent)), 2)

    def rel_pick(self, pick, among):
        """among[choice[pick]].state.p_pos - agent.state.p_pos: the per-world goal (`agent.goal_a.state.p_pos - ...`).
        `among` = the entity list the pick indexes (consecutive in world.entities, e.g. world.landmarks)."""
        base = self._e(among[0])
        for k, e in enumerate(among):
            if self._e(e) != base + k:
                raise _abi.MpeError("spec: the entities a pick chooses among must be consecutive in world.entities")
        return self._emit(_op(_abi.MPE_ROW_OBS_REL_PICK, 0, int(pick), 0, w1=base), 2)

    def comm(self, agent, width=None):
        """agent.state.c (what it said at this step) -- dim_c columns."""
        n = int(self.world.dim_c if width is None else width)
        return self._emit(_op(_abi.MPE_ROW_OBS_COMM, self._a(agent), n), n)

    def const(self, *values):
        """Constants (a colour, zeros for silent agents' utterances ...) -- one column each."""
        for v in values:
            self._emit(_op(_abi.MPE_ROW_OBS_CONST, f0=float(v)), 1)
        return self

    def onehot(self, pick, width, lo=0.0, hi=1.0, offset=0):
        """`width` columns: hi where choice[pick] + offset == column, lo elsewhere -- a goal colour, a key."""
        return self._emit(_op(_abi.MPE_ROW_OBS_ONEHOT, int(pick), int(width), 0, w1=int(offset), f0=lo, f1=hi), int(width))

    def rel_visible(self, agent):
        """rel(agent), zeros when the observer cannot see it (regions of the world: simple_world_comm.py:231-261)."""
        return self._emit(_op(_abi.MPE_ROW_OBS_REL_VIS, self._a(agent)), 2)

    def vel_visible(self, agent):
        return self._emit(_op(_abi.MPE_ROW_OBS_VEL_VIS, self._a(agent)), 2)

    def code(self, width, source_hash):
        """`width` columns written by traced device code (symtrace.py: a reference-style file's observation callback as
        straight-line code appended to the compiled program; `source_hash`: 64 bits of that source, part of the program's
        identity).  Programs with code ops run compiled in only."""
        return self._emit(_code_op(_abi.MPE_ROW_OBS_CODE, int(width), source_hash), int(width))

    def in_region(self, region, ent=None):
        """+1 / -1: is the entity (default: the observer) inside region number `region` -- 1 column."""
        return self._emit(_op(_abi.MPE_ROW_OBS_IN_REGION, self._e(ent, True), int(region)), 1)


class RewardSpec(_Spec):
    """One agent's reward as an ordered list of terms: a value register, two accumulators (acc 0 is the reward), 8 slots.
    Every method returns self; `min_dist(...).add(-1.0)` reads "minus the distance of the nearest ..."."""

    def dist2(self, a, b):
        """v = |a - b|^2"""
        self.ops.append(_op(_abi.MPE_ROW_R_D2, self._e(a), self._e(b)))
        return self

    def min_dist2(self, agents, b):
        """v = min over `agents` of |agent - b|^2 (in list order)"""
        for k, a in enumerate(agents):
            self.ops.append(_op(_abi.MPE_ROW_R_D2 if k == 0 else _abi.MPE_ROW_R_MIN_D2, self._e(a), self._e(b)))
        return self

    def min_dist2_from(self, a, targets):
        """v = min over `targets` of |a - target|^2"""
        for k, t in enumerate(targets):
            self.ops.append(_op(_abi.MPE_ROW_R_D2 if k == 0 else _abi.MPE_ROW_R_MIN_D2, self._e(a), self._e(t)))
        return self

    def dist2_pick(self, a, pick, among, minimum=False):
        """v = |a - among[choice[pick]]|^2 (or min(v, that))"""
        base = self._e(among[0])
        self.ops.append(_op(_abi.MPE_ROW_R_MIN_D2_PICK if minimum else _abi.MPE_ROW_R_D2_PICK, self._e(a), int(pick), 0, w1=base))
        return self

    def sqrt(self):
        self.ops.append(_op(_abi.MPE_ROW_R_SQRT))
        return self

    def min_dist(self, agents, b):
        """v = min over `agents` of |agent - b| (the minimum is taken on the squares: sqrt is monotone)"""
        return self.min_dist2(agents, b).sqrt()

    def dist(self, a, b):
        return self.dist2(a, b).sqrt()

    def bound(self, ent, axis):
        """v = bound(|coordinate|): 0 below 0.9, 10 (x - 0.9) to 1.0, min(exp(2x - 2), 10) beyond (simple_tag.py:103-108)"""
        self.ops.append(_op(_abi.MPE_ROW_R_BOUND, self._e(ent), int(axis)))
        return self

    def comm_error(self, agent, pick):
        """v = sum_c (agent.state.c[c] - onehot(choice[pick])[c])^2, 0 when the utterance is all zeros (simple_crypto.py:97-124)"""
        self.ops.append(_op(_abi.MPE_ROW_R_COMM_ERR, self._a(agent), int(pick)))
        return self

    def comm_sum(self, agent):
        self.ops.append(_op(_abi.MPE_ROW_R_COMM_SUM, self._a(agent)))
        return self

    def value(self, x):
        self.ops.append(_op(_abi.MPE_ROW_R_CONST, f0=float(x)))
        return self

    def save(self, slot):
        self.ops.append(_op(_abi.MPE_ROW_R_SAVE, int(slot)))
        return self

    def load(self, slot):
        self.ops.append(_op(_abi.MPE_ROW_R_LOAD, int(slot)))
        return self

    def zero(self, acc=0):
        self.ops.append(_op(_abi.MPE_ROW_R_ZERO, 0, 0, int(acc)))
        return self

    def add(self, coef=1.0, acc=0):
        """acc = acc + coef * v"""
        self.ops.append(_op(_abi.MPE_ROW_R_ADD, 0, 0, int(acc), f0=float(coef)))
        return self

    def add_if_touching(self, a, b, coef, acc=0):
        """if |a - b| < a.size + b.size (strict, decided exactly): acc = acc + coef -- `if self.is_collision(a, b): rew += coef`"""
        self.ops.append(_op(_abi.MPE_ROW_R_ADD_IF_HIT, self._e(a), self._e(b), int(acc), f0=float(coef)))
        return self

    def add_acc1(self):
        """acc0 = acc0 + acc1 (a sub-sum the reference forms separately)"""
        self.ops.append(_op(_abi.MPE_ROW_R_ADD_ACC))
        return self


def _reward_code(self, source_hash):
    """acc0 = the traced reward (symtrace.py); see ObsSpec.code"""
    self.ops.append(_code_op(_abi.MPE_ROW_R_CODE, 0, source_hash))
    return self


RewardSpec.code = _reward_code


class DoneSpec(RewardSpec):
    """One agent's done condition (the reference's `done_callback(agent, world)`, environment.py:132-135) as tests on the
    reward machine's value register: the value methods of RewardSpec (dist, min_dist, dist2_pick, abs_pos ...) followed by
    `done_if_gt / done_if_lt`, or `done_if_touching`; the agent is done when ANY test fires.  With `auto_reset` the step, the
    tests and the restart of the finished worlds are ONE launch (mpe_step_rows_episode)."""

    def abs_pos(self, ent, axis):
        """v = |coordinate `axis` of `ent`|"""
        self.ops.append(_op(_abi.MPE_ROW_R_ABS_POS, self._e(ent), int(axis)))
        return self

    def done_if_gt(self, threshold):
        self.ops.append(_op(_abi.MPE_ROW_R_DONE_IF_GT, f0=float(threshold)))
        return self

    def done_if_lt(self, threshold):
        self.ops.append(_op(_abi.MPE_ROW_R_DONE_IF_LT, f0=float(threshold)))
        return self

    def code(self, source_hash):
        """done = done or the traced done callback (symtrace.py)"""
        self.ops.append(_code_op(_abi.MPE_ROW_R_DONE_CODE, 0, source_hash))
        return self

    def done_if_touching(self, a, b):
        """|a - b| < a.size + b.size (strict, decided exactly as the contact tests of the rewards)"""
        self.ops.append(_op(_abi.MPE_ROW_R_DONE_IF_HIT, self._e(a), self._e(b)))
        return self

    def outside(self, ent, bound):
        """`ent` left the square |x|, |y| <= bound"""
        return self.abs_pos(ent, 0).done_if_gt(bound).abs_pos(ent, 1).done_if_gt(bound)


class Regions(object):
    """Landmarks that hide what is inside them (at most two) and the agents that see everybody anyway: the visibility rule
    of ObsSpec.rel_visible / vel_visible / in_region (simple_world_comm.py:231-261: the forests, the leader)."""

    def __init__(self, landmarks=(), all_seeing=()):
        self.landmarks = list(landmarks)
        self.all_seeing = list(all_seeing)


# ---- peephole pass: runs of per-entity ops -> range forms (one decode, an inner loop in the kernel) ------------------------------
def _fields(op):
    w0 = op[0] & 0xFFFFFFFF
    return w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255, (w0 >> 24) & 255


_RANGE_OF = {_abi.MPE_ROW_OBS_REL: _abi.MPE_ROW_OBS_REL_RANGE, _abi.MPE_ROW_OBS_VEL: _abi.MPE_ROW_OBS_VEL_RANGE,
             _abi.MPE_ROW_OBS_REL_VIS: _abi.MPE_ROW_OBS_REL_VIS_RANGE, _abi.MPE_ROW_OBS_VEL_VIS: _abi.MPE_ROW_OBS_VEL_VIS_RANGE}


def fuse_obs(ops, me):
    """Consecutive rel / vel (/ visible) ops over consecutive entities -> one range op (a gap exactly at the observing agent is
    the range's skip-self flag); equal constants in a row -> CONST_N.  Same columns, same values, fewer decodes."""
    out, k = [], 0
    while k < len(ops):
        code, a0, a1, a2 = _fields(ops[k])
        if code in _RANGE_OF and a0 != SELF:
            ents, j = [a0], k + 1
            while j < len(ops):
                c2, b0, _, _ = _fields(ops[j])
                nxt = ents[-1] + 1
                if c2 == code and b0 != SELF and (b0 == nxt or (nxt == me and b0 == nxt + 1)):
                    ents.append(b0)
                    j += 1
                else:
                    break
            if len(ents) >= 2:
                first, count = ents[0], ents[-1] - ents[0] + 1
                skip = 1 if count != len(ents) else 0          # the only possible gap is the observer
                out.append(_op(_RANGE_OF[code], first, count, skip))
                k = j
                continue
        if code == _abi.MPE_ROW_OBS_CONST:
            j = k + 1
            while j < len(ops) and _fields(ops[j])[0] == code and ops[j][2] == ops[k][2] and j - k < 255:
                j += 1
            if j - k >= 2:
                out.append((_op(_abi.MPE_ROW_OBS_CONST_N, 0, j - k)[0], 0, ops[k][2], 0))
                k = j
                continue
        out.append(ops[k])
        k += 1
    return out


def fuse_reward(ops):
    """D2 + MIN_D2 chains over consecutive entities -> MIN_D2_RANGE / MIN_D2_TO_RANGE (same order); runs of ADD_IF_HIT with one
    coefficient into one accumulator -> contact grids (every contact adds the same constant, a miss adds nothing: the sum does
    not depend on the order of the tests)."""
    out, k = [], 0
    while k < len(ops):
        code, a0, a1, a2 = _fields(ops[k])
        if code == _abi.MPE_ROW_R_D2:
            j, same_b, same_a = k + 1, True, True
            chain = [(a0, a1)]
            while j < len(ops) and _fields(ops[j])[0] == _abi.MPE_ROW_R_MIN_D2:
                chain.append(_fields(ops[j])[1:3])
                j += 1
            if len(chain) >= 2:
                if all(b == a1 for _, b in chain) and [a for a, _ in chain] == list(range(a0, a0 + len(chain))):
                    out.append(_op(_abi.MPE_ROW_R_MIN_D2_RANGE, a0, a1, 0, w1=len(chain)))
                    k = j
                    continue
                if all(a == a0 for a, _ in chain) and [b for _, b in chain] == list(range(a1, a1 + len(chain))):
                    out.append(_op(_abi.MPE_ROW_R_MIN_D2_TO_RANGE, a0, a1, 0, w1=len(chain)))
                    k = j
                    continue
        if code == _abi.MPE_ROW_R_ADD_IF_HIT:
            j = k + 1
            while j < len(ops) and _fields(ops[j])[0] == code and _fields(ops[j])[3] == a2 and ops[j][2] == ops[k][2]:
                j += 1
            pairs = sorted(set(_fields(o)[1:3] for o in ops[k:j]))
            if len(pairs) == j - k and len(pairs) >= 2:        # no pair twice (a repeated test would add twice: keep those as they are)
                by_b = {}
                for a, b in pairs:
                    by_b.setdefault(b, []).append(a)
                As = sorted(set(a for a, _ in pairs))
                Bs = sorted(by_b)
                full = all(by_b[b] == As for b in Bs) and As == list(range(As[0], As[0] + len(As))) and Bs == list(range(Bs[0], Bs[0] + len(Bs)))
                grids = []
                if full:
                    grids.append((As[0], len(As), Bs[0], len(Bs)))
                else:                                            # per partner b: maximal runs of consecutive a
                    for b in Bs:
                        run = [by_b[b][0]]
                        for a in by_b[b][1:] + [None]:
                            if a is not None and a == run[-1] + 1:
                                run.append(a)
                            else:
                                grids.append((run[0], len(run), b, 1))
                                run = [a]
                for fa, na, fb, nb in grids:
                    if na * nb == 1:
                        out.append((_op(_abi.MPE_ROW_R_ADD_IF_HIT, fa, fb, a2)[0], 0, ops[k][2], 0))
                    else:
                        out.append((_op(_abi.MPE_ROW_R_ADD_IF_HIT_GRID, fa, fb, a2)[0], na | (nb << 8), ops[k][2], 0))
                k = j
                continue
        out.append(ops[k])
        k += 1
    # second pass: [MIN_D2_RANGE(a.., b), SQRT, ADD coef] for consecutive targets b -> one grid op (the nearest-agent term of
    # simple_spread.py:72-77 over all landmarks)
    ops, out, k = out, [], 0
    while k < len(ops):
        def triple(q):      # (q + 2 < len(ops) is the caller's)
            c0, x0, x1, _ = _fields(ops[q])
            if c0 != _abi.MPE_ROW_R_MIN_D2_RANGE or _fields(ops[q + 1])[0] != _abi.MPE_ROW_R_SQRT or _fields(ops[q + 2])[0] != _abi.MPE_ROW_R_ADD:
                return None
            return (x0, ops[q][1], x1, _fields(ops[q + 2])[3], ops[q + 2][2])      # first a, count, b, acc, coef bits
        t0 = triple(k) if k + 2 < len(ops) else None
        if t0 is not None:
            nb, q = 1, k + 3
            while q + 2 < len(ops):
                t = triple(q)
                if t is None or t[0] != t0[0] or t[1] != t0[1] or t[3] != t0[3] or t[4] != t0[4] or t[2] != t0[2] + nb:
                    break
                nb += 1
                q += 3
            if nb >= 2 or t0[1] >= 2:
                out.append((_op(_abi.MPE_ROW_R_ADD_MIN_DIST_GRID, t0[0], t0[2], t0[3])[0], t0[1] | (nb << 8), t0[4], 0))
                k = q
                continue
        out.append(ops[k])
        k += 1
    return out


class RowProgram(object):
    """The compiled programs of one env: ops on the device + the MpeRowProgram header the C ABI takes."""

    def __init__(self, world, obs_specs, reward_specs, regions=None, fuse=None, done_specs=None, source=None, reset_boxes=None,
                 n_shared=0):
        """fuse: run the peephole pass (runs of per-entity ops -> range forms); False keeps one op per spec call (the A/B);
        None: the module's FUSE switch.  done_specs: one DoneSpec (or None: never done) per agent.  source: the device
        functions the program's code ops call (symtrace.hip_source), appended to the generated header of the compiled form.
        reset_boxes: one (lo_x, hi_x, lo_y, hi_y) per entity -- reset_world places entity e uniformly in that box
        (`np.random.uniform(lo, hi, dim_p)`); every restart the library draws for this program (episode ends inside the launch,
        rollouts, mpe_reset_rows) then uses it.  None: the reference's placement (agents [-1,1)^2, landmarks [-r,r)^2)."""
        fuse = FUSE if fuse is None else fuse
        self.source = source
        A = len(world.agents)
        if len(world.entities) > _abi.MPE_ROWS_MAX_ENTITIES:
            raise _abi.MpeError("row programs cover at most %d entities" % _abi.MPE_ROWS_MAX_ENTITIES)
        if len(obs_specs) != A or len(reward_specs) != A:
            raise _abi.MpeError("one ObsSpec and one RewardSpec per agent")
        ops, begin = [], [0]
        for i, o in enumerate(obs_specs):
            ops += fuse_obs(o.ops, i) if fuse else o.ops
            begin.append(len(ops))
        self.widths = [o.width for o in obs_specs]
        rbegin = [len(ops)]
        for i, r in enumerate(reward_specs):         # agent i's reward program: both accumulators start at 0, STORE at the end
            ops.append(_op(_abi.MPE_ROW_R_ZERO, 0, 0, 0))
            ops.append(_op(_abi.MPE_ROW_R_ZERO, 0, 0, 1))
            ops += fuse_reward(r.ops) if fuse else r.ops
            ops.append(_op(_abi.MPE_ROW_R_STORE, i))
            rbegin.append(len(ops))
        dbegin = [0] * (A + 1)
        self.has_done = bool(done_specs) and any(d is not None and d.ops for d in done_specs)
        if self.has_done:
            if len(done_specs) != A:
                raise _abi.MpeError("one DoneSpec (or None) per agent")
            dbegin = [len(ops)]
            for d in done_specs:
                ops += (fuse_reward(d.ops) if fuse else d.ops) if d is not None else []
                dbegin.append(len(ops))
        self.n_ops = len(ops)
        flat = [w for op in ops for w in op]
        self.ops_host = (C.c_int32 * max(1, len(flat)))(*flat)
        self.ops_device = torch.tensor(flat if flat else [0], dtype=torch.int32, device=world.device)
        self.header_device = torch.zeros(_abi.MPE_ROWS_HEADER_BYTES // 4, dtype=torch.int32, device=world.device)   # the library's
        p = _abi.MpeRowProgram()
        p.ops_device = self.ops_device.data_ptr()
        p.header_device = self.header_device.data_ptr()
        p.header_hash = 0
        p.n_ops = self.n_ops
        for i in range(_abi.MPE_ROWS_MAX_ENTITIES + 1):
            p.obs_begin[i] = begin[min(i, A)]
        for i in range(_abi.MPE_ROWS_MAX_ENTITIES + 1):
            p.rew_begin[i] = rbegin[min(i, A)]
        for i in range(_abi.MPE_ROWS_MAX_ENTITIES + 1):
            p.done_begin[i] = dbegin[min(i, A)]
        p.n_vel = world.n_dynamic
        regions = regions or Regions()
        if len(regions.landmarks) > 2:
            raise _abi.MpeError("at most two regions")
        p.n_regions = len(regions.landmarks)
        ents = world.entities
        for r, lm in enumerate(regions.landmarks):
            p.region_entity[r] = lm if isinstance(lm, int) else next(k for k, e in enumerate(ents) if e is lm)
        mask = 0
        for a in regions.all_seeing:
            mask |= 1 << (a if isinstance(a, int) else next(k for k, e in enumerate(world.agents) if e is a))
        p.all_seeing = mask
        p.reset_boxes = 0
        if reset_boxes is not None:
            if len(reset_boxes) != len(world.entities):
                raise _abi.MpeError("reset_boxes: one (lo_x, hi_x, lo_y, hi_y) per entity")
            p.reset_boxes = 1
            for e, (lx, hx, ly, hy) in enumerate(reset_boxes):
                if not (hx >= lx and hy >= ly):
                    raise _abi.MpeError("reset_boxes: entity %d has an empty box" % e)
                p.reset_box[e][0], p.reset_box[e][1], p.reset_box[e][2], p.reset_box[e][3] = lx, hx - lx, ly, hy - ly
        codes = (_abi.MPE_ROW_OBS_CODE, _abi.MPE_ROW_R_CODE, _abi.MPE_ROW_R_DONE_CODE)
        p.traced = 1 if any((op[0] & 0xFF) in codes for op in ops) else 0
        if p.traced and not source:
            raise _abi.MpeError("a program with code ops needs the source of the functions they call")
        self.traced = bool(p.traced)
        p.n_shared = int(n_shared) if p.traced else 0      # (values the traced source's traced_shared computes once per world)
        self.struct = p
        self.ref = C.byref(p)

    def validate(self, desc):
        _abi.check(_abi.lib().mpe_rows_validate(C.byref(desc), self.ref, self.ops_host), "mpe_rows_validate")

    # ---- the program compiled in (include/mpe_hip.h: mpe_rows_static_source / mpe_rows_load_image) ----------------------------
    def static_source(self, desc):
        """The generated header (name, dims, tables, ops as #defines) csrc/mpe_rows.hip is compiled with."""
        need = C.c_size_t(0)
        L = _abi.lib()
        _abi.check(L.mpe_rows_static_source(C.byref(desc), self.ref, self.ops_host, None, 0, C.byref(need)), "mpe_rows_static_source")
        buf = C.create_string_buffer(need.value)
        _abi.check(L.mpe_rows_static_source(C.byref(desc), self.ref, self.ops_host, buf, need.value, None), "mpe_rows_static_source")
        return buf.value.decode() + (self.source or "")

    def compile(self, desc, verbose=False, cached_only=False):
        """Compile the program in for THIS descriptor (hipcc --genco, cached by content under lib/rows_cache/) and attach it:
        mpe_rows / mpe_step_rows / mpe_episode_finish launch the image while the descriptor stays what it was compiled for.
        cached_only: attach an image the cache already holds, never run hipcc; returns 0 when there is none."""
        from . import _build
        src = self.static_source(desc)
        image = _build.cached_rows_image(src) if cached_only else _build.compile_rows_image(src, verbose=verbose)
        if image is None:
            return 0
        self._image = C.create_string_buffer(image, len(image))
        _abi.check(_abi.lib().mpe_rows_load_image(C.byref(desc), self.ref, self.ops_host, self._image, len(image)), "mpe_rows_load_image")
        return len(image)

    def unload(self):
        _abi.lib().mpe_rows_unload_image(self.ref)
        self._image = None

    def image_active(self, desc):
        return bool(_abi.lib().mpe_rows_image_active(C.byref(desc), self.ref))

    def __del__(self):
        try:
            if self.struct.image:
                _abi.lib().mpe_rows_unload_image(self.ref)
        except Exception:
            pass


def compile_scenario(scenario, world):
    """-> RowProgram when the scenario describes every agent's observation AND reward as specs (`obs_spec(agent, world)`,
    `reward_spec(agent, world)`, optional `regions(world)`), else None."""
    osf, rsf = getattr(scenario, "obs_spec", None), getattr(scenario, "reward_spec", None)
    if osf is None or rsf is None:
        return None
    obs = [osf(a, world) for a in world.agents]
    rew = [rsf(a, world) for a in world.agents]
    if any(o is None for o in obs) or any(r is None for r in rew):
        return None
    rg = scenario.regions(world) if hasattr(scenario, "regions") else None
    dsf = getattr(scenario, "done_spec", None)
    done = [dsf(a, world) for a in world.agents] if dsf is not None else None
    src = scenario.row_source(world) if hasattr(scenario, "row_source") else None
    boxes = scenario.reset_boxes(world) if callable(getattr(scenario, "reset_boxes", None)) else None
    n_shared = scenario.row_shared(world) if callable(getattr(scenario, "row_shared", None)) else 0
    return RowProgram(world, obs, rew, rg, done_specs=done, source=src, reset_boxes=boxes, n_shared=n_shared)


# built-in scenarios whose callbacks are written for any team size: where no fused kernel exists for a shape, the env runs
# their specs (World.step + mpe_rows) instead of falling back to the torch callbacks
BUILTIN_PROGRAM_KINDS = {_abi.MPE_SCN_SIMPLE: "simple", _abi.MPE_SCN_ADVERSARY: "simple_adversary", _abi.MPE_SCN_PUSH: "simple_push",
                         _abi.MPE_SCN_WORLD_COMM: "simple_world_comm"}


def builtin_program(name, world):
    obs, rew, regions = builtin_specs(name, world)
    return RowProgram(world, obs, rew, regions)


# ---- the nine shipped scenarios as specs: each follows its fused kernel's (= the reference callback's) order ---------------------
def _others(world, agent):
    return [a for a in world.agents if a is not agent]


def builtin_specs(name, world):
    """(obs_specs, reward_specs, regions) of a built-in scenario's world (as its make_world builds it)."""
    ag, lm = world.agents, world.landmarks
    A = len(ag)
    nadv = sum(1 for a in ag if getattr(a, "adversary", False))
    obs, rew, regions = [], [], None
    collide = lambda a: bool(a.collide)
    for i, me in enumerate(ag):
        o, r = ObsSpec(world, me), RewardSpec(world, me)
        adv = bool(getattr(me, "adversary", False))
        good = [a for a in ag if not getattr(a, "adversary", False)]
        advs = [a for a in ag if getattr(a, "adversary", False)]
        if name == "simple":                       # simple.py:41-50
            o.vel()
            for l in lm:
                o.rel(l)
            r.dist2(me, lm[0]).add(-1.0)
        elif name == "simple_spread":              # simple_spread.py:72-100
            o.vel().pos()
            for l in lm:
                o.rel(l)
            for a in _others(world, me):
                o.rel(a)
            o.const(*([0.0] * (world.dim_c * (A - 1))))
            for l in lm:
                r.min_dist(ag, l).add(-1.0)
            if collide(me):
                r.add_if_touching(me, me, -1.0)             # the agent against itself (Q1)
                for a in _others(world, me):
                    r.add_if_touching(a, me, -1.0)
        elif name == "simple_tag":                 # simple_tag.py:84-147
            o.vel().pos()
            for l in lm:
                o.rel(l)
            for a in _others(world, me):
                o.rel(a)
            for a in _others(world, me):
                if not a.adversary:
                    o.vel(a)
            if adv:
                if collide(me):
                    for g in good:
                        for v in advs:
                            r.add_if_touching(g, v, 10.0)
            else:
                if collide(me):
                    for v in advs:
                        r.add_if_touching(me, v, -10.0)
                r.bound(me, 0).add(-1.0).bound(me, 1).add(-1.0)
        elif name == "simple_adversary":           # simple_adversary.py:76-139
            if not adv:
                o.rel_pick(0, lm)
            for l in lm:
                o.rel(l)
            for a in _others(world, me):
                o.rel(a)
            if adv:
                r.dist2_pick(me, 0, lm).add(-1.0)
            else:
                for k, g in enumerate(good):
                    r.dist2_pick(g, 0, lm, minimum=k > 0)
                r.sqrt().add(-1.0)
                for v in advs:
                    r.dist2_pick(v, 0, lm).sqrt().add(1.0, acc=1)
                r.add_acc1()
        elif name == "simple_push":                # simple_push.py:60-96
            o.vel()
            if not adv:
                o.rel_pick(0, lm)
                o.onehot(0, 3, 0.25, 0.75, offset=1)
            for l in lm:
                o.rel(l)
            if not adv:
                for k in range(len(lm)):
                    o.const(*[(0.1 + 0.8) if k + 1 == c else 0.1 for c in range(3)])
            for a in _others(world, me):
                o.rel(a)
            if adv:
                for k, g in enumerate(good):
                    r.dist2_pick(g, 0, lm, minimum=k > 0)
                r.sqrt().add(1.0)
                r.dist2_pick(me, 0, lm).sqrt().add(-1.0)
            else:
                r.dist2_pick(me, 0, lm).sqrt().add(-1.0)
        elif name == "simple_speaker_listener":    # simple_speaker_listener.py:63-92
            if i == 0:
                o.onehot(0, 3, 0.15, 0.65)
            else:
                o.vel()
                for l in lm:
                    o.rel(l)
                o.comm(ag[0])
            r.dist2_pick(ag[1], 0, lm).add(-1.0)
        elif name == "simple_reference":           # simple_reference.py:57-83
            o.vel()
            for l in lm:
                o.rel(l)
            o.onehot(i, 3, 0.25, 0.75)
            o.comm(ag[1 - i])
            r.dist2_pick(ag[1 - i], i, lm).add(-1.0)
        elif name == "simple_crypto":              # simple_crypto.py:97-169 (goal = pick 0, key = pick 1)
            dc = world.dim_c
            if i == 0:
                o.comm(ag[2])
                r.comm_error(ag[0], 0).add(-1.0)
            else:
                if i == 1:
                    o.onehot(1, dc).comm(ag[2])
                else:
                    o.onehot(0, dc).onehot(1, dc)
                r.comm_error(ag[0], 0).add(1.0)
                r.comm_error(ag[1], 0).add(-1.0, acc=1)
                r.add_acc1()
        elif name == "simple_world_comm":          # simple_world_comm.py:143-289: landmarks = [obstacle] + food (2) + forests (2)
            regions = Regions(lm[3:5], [ag[0]])
            food = lm[1:3]
            o.vel().pos()
            for l in lm:
                o.rel(l)
            for a in _others(world, me):
                o.rel_visible(a)
            if not adv:
                o.in_region(0).in_region(1)
            for a in good:
                if a is not me:
                    o.vel_visible(a)
            if adv:
                o.in_region(0).in_region(1)
                o.comm(ag[0])
                for k, g in enumerate(good):
                    r.ops.append(_op(_abi.MPE_ROW_R_D2 if k == 0 else _abi.MPE_ROW_R_MIN_D2, r._e(g), r._e(me)))
                r.sqrt().add(-0.1)
                if collide(me):
                    for g in good:
                        for v in advs:
                            r.add_if_touching(g, v, 5.0)
            else:
                if collide(me):
                    for v in advs:
                        r.add_if_touching(me, v, -5.0)
                r.bound(me, 0).add(-2.0).bound(me, 1).add(-2.0)
                for f in food:
                    r.add_if_touching(me, f, 2.0)
                r.min_dist2_from(me, food).sqrt().add(0.05)
        else:
            raise KeyError(name)
        obs.append(o)
        rew.append(r)
    return obs, rew, regions
