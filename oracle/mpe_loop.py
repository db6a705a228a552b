"""Per-object fp64 loop oracle: one world, one entity pair at a time (TEST INFRASTRUCTURE).

Same arithmetic *and the same cost structure* as the reference: a Python loop over entities /
entity pairs with tiny NumPy operations inside, float64 throughout.  It is what `bench.py`
times as `cpu_baseline` (kind "port") because `/root/reference` does not exist on the GPU box.
It is pinned against the real reference by tests/test_oracle_golden.py.

Reference lines restated (operation order kept):
  decode      multiagent/environment.py:144-181   u = (a1-a2, a3-a4) * (accel or 5.0)
  forces      multiagent/core.py:134-155,180-196  action force, then a<b pair loop, soft contact
  integrate   multiagent/core.py:158-169          damp, += f/m*dt, speed clamp, += v*dt
  obs/reward  scenarios/simple.py:41-50, simple_spread.py:47-100, simple_tag.py:57-147
  gather      multiagent/environment.py:92-102    per-agent lists, shared reward = np.sum
  reset       scenarios/*: reset_world            agents then landmarks, uniform(-1,1) (tag lm .9)
"""
import numpy as np


class LoopEnv(object):
    def __init__(self, spec, benchmark=False):
        self.spec = spec
        self.benchmark = benchmark
        self.n = spec.n_agents
        E = spec.n_entities
        self.pos = [np.zeros(2) for _ in range(E)]
        self.vel = [np.zeros(2) for _ in range(E)]
        self.comm = [np.zeros(spec.dim_c) for _ in range(spec.n_agents)]
        self.reset()  # the reference's make_world ends with reset_world (simple_spread.py:28)

    # ---- reset: draws from the process-global np.random in the reference's order -------------
    def reset(self):
        s = self.spec
        for i in range(s.n_agents):
            self.pos[i] = np.random.uniform(-1, +1, 2)
            self.vel[i] = np.zeros(2)
            self.comm[i] = np.zeros(s.dim_c)
        r = s.landmark_range
        for k in range(s.n_agents, s.n_entities):
            self.pos[k] = np.random.uniform(-r, +r, 2)
            self.vel[k] = np.zeros(2)
        return [self._observe(i) for i in range(self.n)]

    def set_state(self, pos, vel):
        for e in range(self.spec.n_entities):
            self.pos[e] = np.array(pos[e], dtype=np.float64)
        for i in range(self.spec.n_agents):
            self.vel[i] = np.array(vel[i], dtype=np.float64)

    # ---- one step -----------------------------------------------------------------------------
    def step(self, action_n):
        s = self.spec
        E = s.n_entities
        force = [None] * E
        # decode + action force
        for i in range(s.n_agents):
            a = action_n[i]
            u = np.zeros(2)
            u[0] += a[1] - a[2]
            u[1] += a[3] - a[4]
            u *= (5.0 if s.accel[i] is None else s.accel[i])
            if s.movable[i]:
                force[i] = u + 0.0
        # pairwise soft contact
        k = s.contact_margin
        for a in range(E):
            for b in range(a + 1, E):
                if not (s.collide[a] and s.collide[b]):
                    continue
                delta = self.pos[a] - self.pos[b]
                dist = np.sqrt(np.sum(np.square(delta)))
                dist_min = s.size[a] + s.size[b]
                pen = np.logaddexp(0, -(dist - dist_min) / k) * k
                f = s.contact_force * delta / dist * pen
                if s.movable[a]:
                    force[a] = f + (0.0 if force[a] is None else force[a])
                if s.movable[b]:
                    force[b] = -f + (0.0 if force[b] is None else force[b])
        # integrate
        for e in range(E):
            if not s.movable[e]:
                continue
            v = self.vel[e] * (1 - s.damping)
            if force[e] is not None:
                v += (force[e] / s.mass_of(e)) * s.dt
            ms = s.max_speed[e] if e < s.n_agents else None
            if ms is not None:
                speed = np.sqrt(np.square(v[0]) + np.square(v[1]))
                if speed > ms:
                    v = v / np.sqrt(np.square(v[0]) + np.square(v[1])) * ms
            self.vel[e] = v
            self.pos[e] = self.pos[e] + v * s.dt
        for i in range(s.n_agents):
            self.comm[i] = np.zeros(s.dim_c)  # every in-scope agent is silent (core.py:173-174)
        # gather
        obs_n, rew_n, done_n, info = [], [], [], {"n": []}
        for i in range(self.n):
            obs_n.append(self._observe(i))
            rew_n.append(self._reward(i))
            done_n.append(False)
            info["n"].append(self._bench(i) if self.benchmark else {})
        total = np.sum(rew_n)
        if s.collaborative:
            rew_n = [total] * self.n
        return obs_n, rew_n, done_n, info

    # ---- scenario pieces ------------------------------------------------------------------------
    def _dist(self, a, b):
        return np.sqrt(np.sum(np.square(self.pos[a] - self.pos[b])))

    def _touch(self, a, b):
        return bool(self._dist(a, b) < self.spec.size[a] + self.spec.size[b])

    def _observe(self, i):
        s = self.spec
        A = s.n_agents
        lm = [self.pos[l] - self.pos[i] for l in range(A, s.n_entities)]
        if s.name == "simple":
            return np.concatenate([self.vel[i]] + lm)
        others = [self.pos[j] - self.pos[i] for j in range(A) if j != i]
        if s.name == "simple_spread":
            comm = [self.comm[j] for j in range(A) if j != i]
            return np.concatenate([self.vel[i], self.pos[i]] + lm + others + comm)
        if s.name == "simple_tag":
            gv = [self.vel[j] for j in range(A) if j != i and not s.adversary[j]]
            return np.concatenate([self.vel[i], self.pos[i]] + lm + others + gv)
        raise KeyError(s.name)

    def _reward(self, i):
        s = self.spec
        A = s.n_agents
        if s.name == "simple":
            return -np.sum(np.square(self.pos[i] - self.pos[A]))
        if s.name == "simple_spread":
            rew = 0
            for l in range(A, s.n_entities):
                rew -= min([self._dist(a, l) for a in range(A)])
            if s.collide[i]:
                for a in range(A):
                    if self._touch(a, i):
                        rew -= 1
            return rew
        if s.name == "simple_tag":
            advs = [j for j in range(A) if s.adversary[j]]
            good = [j for j in range(A) if not s.adversary[j]]
            rew = 0
            if s.adversary[i]:
                if s.collide[i]:                       # simple_tag.py:124
                    for g in good:
                        for a in advs:
                            if self._touch(g, a):
                                rew += 10
                return rew
            if s.collide[i]:                           # simple_tag.py:97
                for a in advs:
                    if self._touch(a, i):
                        rew -= 10
            for p in range(2):
                x = abs(self.pos[i][p])
                if x < 0.9:
                    pass
                elif x < 1.0:
                    rew -= (x - 0.9) * 10
                else:
                    rew -= min(np.exp(2 * x - 2), 10)
            return rew
        raise KeyError(s.name)

    def _bench(self, i):
        s = self.spec
        A = s.n_agents
        if s.name == "simple_spread":
            rew, hits, occupied, md = 0, 0, 0, 0
            for l in range(A, s.n_entities):
                m = min([self._dist(a, l) for a in range(A)])
                md += m
                rew -= m
                if m < 0.1:
                    occupied += 1
            for a in range(A):
                if self._touch(a, i):
                    rew -= 1
                    hits += 1
            return (rew, hits, md, occupied)
        if s.name == "simple_tag":
            if not s.adversary[i]:
                return 0
            return sum(1 for g in range(A) if not s.adversary[g] and self._touch(g, i))
        return {}
