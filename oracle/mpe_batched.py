"""Batched NumPy oracle: the reference's arithmetic vectorised over B worlds (TEST INFRASTRUCTURE).

`dtype=np.float64` is the scalable truth the HIP path is compared with (<=1e-5, teacher-forced);
`dtype=np.float32` repeats the same operation order in single precision and is the oracle for the
bit-exact integer outputs (collision counts, occupied landmarks), see DESIGN.md "parity protocol".
Operation order follows SURVEY.md appendix A.1; each block cites the reference lines it restates.
Checked against the real reference by tests/test_oracle_golden.py (<=1e-12 in fp64).

Layouts (oracle-internal, chosen for NumPy, NOT the device layout):
  pos [B, E, 2]   vel [B, A, 2]   actions [A, B, 5]   obs: list of A arrays [B, D_i]
  rew [A, B]      done [A, B] bool
"""
import numpy as np


def seeded_initial_state(spec, seeds):
    """Initial states exactly as `np.random.seed(s); env.reset()` would draw them per world
    (reset_world order: all agents, then all landmarks, two uniforms each --
    simple_spread.py:39-45, simple_tag.py:47-54, simple.py:33-39)."""
    B, E, A = len(seeds), spec.n_entities, spec.n_agents
    pos = np.zeros((B, E, 2))
    for b, s in enumerate(seeds):
        rs = np.random.RandomState(int(s))  # same MT19937 stream as the global np.random.seed(s)
        for e in range(E):
            r = 1.0 if e < A else spec.landmark_range
            pos[b, e] = rs.uniform(-r, +r, 2)
    return pos, np.zeros((B, A, 2))


class BatchedOracle(object):
    def __init__(self, spec, batch, dtype=np.float64, benchmark=False):
        self.spec, self.B, self.dt_, self.benchmark = spec, batch, np.dtype(dtype), benchmark
        self.pos = np.zeros((batch, spec.n_entities, 2), self.dt_)
        # core.py:158-169 integrates EVERY movable entity: agents, and any landmark a scenario makes movable.  vel holds
        # the agents' rows [B,A,2] as everywhere else, or all entities' [B,E,2] when a landmark moves.
        self.n_dyn = spec.n_entities if any(spec.movable[spec.n_agents:]) else spec.n_agents
        self.vel = np.zeros((batch, self.n_dyn, 2), self.dt_)
        self.size = np.asarray(spec.size, self.dt_)

    def set_state(self, pos, vel):
        self.pos = np.array(pos, dtype=self.dt_)
        self.vel = np.array(vel, dtype=self.dt_)

    # ------------------------------------------------------------------ physics (core.py:117-196)
    def decode(self, actions):
        """environment.py:174-181: u = (a1-a2, a3-a4) * (accel or 5.0) -> [A, B, 2]."""
        s = self.spec
        a = np.asarray(actions, self.dt_)
        u = np.stack([a[..., 1] - a[..., 2], a[..., 3] - a[..., 4]], axis=-1)
        sens = np.asarray([5.0 if x is None else x for x in s.accel], self.dt_)
        return u * sens[:, None, None]

    def decode_ids(self, ids):
        """environment.py:161-167 (discrete_action_input): 1:-x 2:+x 3:-y 4:+y (sign quirk Q3)."""
        s = self.spec
        ids = np.asarray(ids)
        u = np.zeros(ids.shape + (2,), self.dt_)
        u[..., 0] = np.where(ids == 1, -1.0, np.where(ids == 2, 1.0, 0.0))
        u[..., 1] = np.where(ids == 3, -1.0, np.where(ids == 4, 1.0, 0.0))
        sens = np.asarray([5.0 if x is None else x for x in s.accel], self.dt_)
        return u * sens[:, None, None]

    def forces(self, u):
        """core.py:134-155,180-196 -> per-entity force [E] list of [B,2] (None = no force)."""
        s = self.spec
        E = s.n_entities
        f = [None] * E
        for i in range(s.n_agents):
            if s.movable[i]:
                f[i] = u[i] + 0.0
        k = self.dt_.type(s.contact_margin)
        C = self.dt_.type(s.contact_force)
        for a in range(E):
            if not s.collide[a]:
                continue
            for b in range(a + 1, E):
                if not s.collide[b]:
                    continue
                if not (s.movable[a] or s.movable[b]):
                    continue  # evaluated but applied to nobody in the reference (Q8)
                delta = self.pos[:, a] - self.pos[:, b]
                dist = np.sqrt(np.square(delta[:, 0]) + np.square(delta[:, 1]))
                dist_min = self.size[a] + self.size[b]
                with np.errstate(all="ignore"):
                    pen = np.logaddexp(self.dt_.type(0), -(dist - dist_min) / k) * k
                    fab = C * delta / dist[:, None] * pen[:, None]
                if s.movable[a]:
                    f[a] = fab + (0.0 if f[a] is None else f[a])
                if s.movable[b]:
                    f[b] = -fab + (0.0 if f[b] is None else f[b])
        return f

    def integrate(self, f):
        """core.py:158-169."""
        s = self.spec
        one_minus = self.dt_.type(1 - s.damping)
        dt = self.dt_.type(s.dt)
        for i in range(self.n_dyn):
            if not s.movable[i]:
                continue
            v = self.vel[:, i] * one_minus
            if f[i] is not None:
                v = v + (f[i] / self.dt_.type(s.mass_of(i))) * dt            # core.py:162
            ms_i = s.max_speed[i] if i < len(s.max_speed) else None          # (per agent; a landmark's Entity.max_speed is None)
            if ms_i is not None:
                ms = self.dt_.type(ms_i)
                speed = np.sqrt(np.square(v[:, 0]) + np.square(v[:, 1]))
                with np.errstate(all="ignore"):
                    clamped = v / speed[:, None] * ms
                v = np.where((speed > ms)[:, None], clamped, v)
            self.vel[:, i] = v
            self.pos[:, i] = self.pos[:, i] + v * dt

    # ------------------------------------------------------------------ scenario outputs
    def _pairdist(self, a_idx, b_idx):
        """[B, len(a), len(b)] distances sqrt(dx^2+dy^2), delta = pos[a]-pos[b]."""
        d = self.pos[:, a_idx, None, :] - self.pos[:, None, b_idx, :]
        return np.sqrt(np.square(d[..., 0]) + np.square(d[..., 1]))

    def observe(self):
        s = self.spec
        A, E = s.n_agents, s.n_entities
        out = []
        for i in range(A):
            me = self.pos[:, i]
            lm = [self.pos[:, l] - me for l in range(A, E)]
            if s.name == "simple":
                cols = [self.vel[:, i]] + lm
            else:
                others = [self.pos[:, j] - me for j in range(A) if j != i]
                if s.name == "simple_spread":
                    tail = [np.zeros((self.B, s.dim_c), self.dt_) for j in range(A) if j != i]
                else:
                    tail = [self.vel[:, j] for j in range(A) if j != i and not s.adversary[j]]
                cols = [self.vel[:, i], me] + lm + others + tail
            out.append(np.concatenate(cols, axis=1))
        return out

    def outputs(self):
        """obs list, rew [A,B], done [A,B], info dict of arrays (benchmark_data)."""
        s = self.spec
        A, E, B = s.n_agents, s.n_entities, self.B
        ag, lms = list(range(A)), list(range(A, E))
        obs = self.observe()
        info = {}
        if s.name == "simple":
            d = self.pos[:, 0] - self.pos[:, A]
            rew = -(np.square(d[:, 0]) + np.square(d[:, 1]))[None, :]
        elif s.name == "simple_spread":
            dal = self._pairdist(ag, lms)                     # [B, A, L]
            mins = dal.min(axis=1)                            # [B, L]  min over agents
            daa = self._pairdist(ag, ag)                      # [B, A(a), A(i)]
            dmin = self.size[ag][:, None] + self.size[ag][None, :]
            hit = daa < dmin[None]                            # includes a == i (Q1)
            counts = hit.sum(axis=1).astype(np.int32)         # [B, A]
            for i in range(A):                                # `if agent.collide:` simple_spread.py:78 (reward), :58 (benchmark_data)
                if not s.collide[i]:
                    counts[:, i] = 0
            lm_term = np.zeros(B, self.dt_)
            for l in range(len(lms)):                         # rew -= min(dists), landmark order
                lm_term = lm_term - mins[:, l]
            per_agent = np.empty((A, B), self.dt_)
            for i in range(A):
                r = lm_term.copy()
                for c in range(A):                            # rew -= 1 once per colliding agent
                    r = r - (counts[:, i] > c).astype(self.dt_)
                per_agent[i] = r
            total = np.sum(np.ascontiguousarray(per_agent.T), axis=1)   # np.sum(reward_n) per world
            rew = np.broadcast_to(total, (A, B)).copy()
            md = np.zeros(B, self.dt_)
            for l in range(len(lms)):
                md = md + mins[:, l]
            info = {"rew": per_agent, "collisions": counts.T.copy(),
                    "min_dists": np.broadcast_to(md, (A, B)).copy(),
                    "occupied_landmarks": np.broadcast_to(
                        (mins < 0.1).sum(axis=1).astype(np.int32), (A, B)).copy()}
        elif s.name == "simple_tag":
            advs = [j for j in ag if s.adversary[j]]
            good = [j for j in ag if not s.adversary[j]]
            dga = self._pairdist(good, advs)                  # [B, G, V]  is_collision(good, adv)
            dmin = self.size[good][:, None] + self.size[advs][None, :]
            hit = dga < dmin[None]
            rew = np.zeros((A, B), self.dt_)
            adv_rew = np.zeros(B, self.dt_)
            for g in range(len(good)):
                for a in range(len(advs)):
                    adv_rew = adv_rew + np.where(hit[:, g, a], 10, 0).astype(self.dt_)
            coll = np.zeros((A, B), np.int32)
            for vi, j in enumerate(advs):
                rew[j] = adv_rew if s.collide[j] else 0.0     # `if agent.collide:` simple_tag.py:124
                coll[j] = hit[:, :, vi].sum(axis=1)           # benchmark_data :57-66 has no such gate
            for gi, j in enumerate(good):
                r = np.zeros(B, self.dt_)
                for a in range(len(advs)):
                    if s.collide[j]:                          # simple_tag.py:97
                        r = r - np.where(hit[:, gi, a], 10, 0).astype(self.dt_)
                for p in range(2):
                    x = np.abs(self.pos[:, j, p])
                    with np.errstate(over="ignore"):
                        far = np.minimum(np.exp(2 * x - 2), 10)
                    r = r - np.where(x < 0.9, 0, np.where(x < 1.0, (x - 0.9) * 10, far))
                rew[j] = r
            info = {"collisions": coll}
        else:
            raise KeyError(s.name)
        done = np.zeros((A, B), bool)
        return obs, rew, done, info

    def step(self, actions=None, ids=None):
        u = self.decode(actions) if ids is None else self.decode_ids(ids)
        self.integrate(self.forces(u))
        return self.outputs()
