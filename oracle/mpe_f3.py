"""Batched NumPy oracle of the six scenarios outside BASELINE.json's configs (TEST INFRASTRUCTURE -- oracle/__init__.py).

The reference's arithmetic vectorised over B worlds, fp64 by default (fp32 repeats the same operation order in
single precision: the oracle of the strict-< outputs).  Physics is `BatchedOracle`'s (core.py:117-196); this file
restates what the six scenarios and the communication half of `_set_action` / `update_agent_state` add:

  _set_action            multiagent/environment.py:144-192  (MultiDiscrete split :148-155, move :161-181, speak :183-190)
  update_agent_state     multiagent/core.py:171-177         (silent -> zeros, else Action.c; c_noise is None everywhere)
  simple_adversary       multiagent/scenarios/simple_adversary.py:57-139
  simple_push            multiagent/scenarios/simple_push.py:34-96
  simple_speaker_listener multiagent/scenarios/simple_speaker_listener.py:34-92
  simple_reference       multiagent/scenarios/simple_reference.py:26-83
  simple_crypto          multiagent/scenarios/simple_crypto.py:50-169
  simple_world_comm      multiagent/scenarios/simple_world_comm.py:89-289

Pinned to the reference by tests/test_oracle_golden.py (tests/golden/f3_*.npz, f3c_*.npz at <= 1e-12) and, in the
build container, by tests/test_oracle_live_reference.py on fresh worlds.

Layouts (oracle-internal): pos [B,E,2]  vel [B,A,2]  c [A,B,dim_c]  choice [B,K] (landmark indices)
actions: list of A arrays [B, d_i], d_i = 5 (moves), dim_c (speaks) or 5 + dim_c (both) -- the rows env.step takes.
"""
import numpy as np

from .mpe_batched import BatchedOracle


def seeded_initial_state_f3(spec, seeds):
    """(pos, vel, choice) exactly as `np.random.seed(s); env.reset()` draws them per world: first the
    np.random.choice picks (goal / key landmarks), then two uniforms per agent, then per landmark
    (simple_adversary.py:44-54, simple_push.py:41-57, simple_speaker_listener.py:40-56, simple_reference.py:33-55,
    simple_crypto.py:63-77).  simple_world_comm.py:101-116 places every landmark, then the food again, then the
    forests again: three loops over the same stream."""
    B, E, A = len(seeds), spec.n_entities, spec.n_agents
    pos = np.zeros((B, E, 2))
    choice = np.zeros((B, len(spec.choice_pops)), np.int64)
    for b, s in enumerate(seeds):
        rs = np.random.RandomState(int(s))
        for k, n in enumerate(spec.choice_pops):
            choice[b, k] = rs.choice(n)
        for i in range(A):
            pos[b, i] = rs.uniform(-1, +1, 2)
        r = spec.landmark_range
        for l in range(A, E):
            pos[b, l] = rs.uniform(-r, +r, 2)
        if spec.name == "simple_world_comm":
            for l in (A + 1, A + 2):      # world.food
                pos[b, l] = rs.uniform(-r, +r, 2)
            for l in (A + 3, A + 4):      # world.forests
                pos[b, l] = rs.uniform(-r, +r, 2)
    return pos, np.zeros((B, A, 2)), choice


class F3Oracle(BatchedOracle):
    def __init__(self, spec, batch, dtype=np.float64, benchmark=False):
        super(F3Oracle, self).__init__(spec, batch, dtype, benchmark)
        self.c = np.zeros((spec.n_agents, batch, spec.dim_c), self.dt_)
        self.choice = np.zeros((batch, len(spec.choice_pops)), np.int64)

    def set_choice(self, choice):
        self.choice = np.asarray(choice, np.int64).reshape(self.B, -1)

    def set_comm(self, c):
        self.c = np.array(c, dtype=self.dt_)

    # ---------------------------------------------------------------- _set_action + World.step
    def split_actions(self, actions):
        """environment.py:144-192 -> (u [A,B,2] with sensitivity applied, words [A] of [B,dim_c] or None)."""
        s = self.spec
        A = s.n_agents
        u = np.zeros((A, self.B, 2), self.dt_)
        words = [None] * A
        for i in range(A):
            a = np.asarray(actions[i], self.dt_)
            k = 0
            if s.movable[i]:
                sens = self.dt_.type(5.0 if s.accel[i] is None else s.accel[i])
                u[i, :, 0] = (a[:, 1] - a[:, 2]) * sens
                u[i, :, 1] = (a[:, 3] - a[:, 4]) * sens
                k = 5
            if not s.silent_of(i):
                words[i] = a[:, k:k + s.dim_c]
        return u, words

    def step(self, actions):
        s = self.spec
        u, words = self.split_actions(actions)
        self.integrate(self.forces(u))
        for i in range(s.n_agents):       # update_agent_state, core.py:171-177
            self.c[i] = 0.0 if s.silent_of(i) else words[i]
        return self.outputs()

    # ---------------------------------------------------------------- helpers
    def _d2(self, a, b):
        """np.sum(np.square(a - b)) of 2-vectors, [B]."""
        d = a - b
        return np.square(d[:, 0]) + np.square(d[:, 1])

    def _dist(self, a, b):
        return np.sqrt(self._d2(a, b))

    def _goal_pos(self, k=0):
        A = self.spec.n_agents
        return self.pos[np.arange(self.B), A + self.choice[:, k]]

    def _onehot(self, idx, n, hot, cold):
        out = np.full((self.B, n), cold, self.dt_)
        out[np.arange(self.B), idx] = hot
        return out

    def _hit(self, a, b):
        """is_collision(a, b): dist < size_a + size_b (strict), entity indices a, b -> [B] bool."""
        return self._dist(self.pos[:, a], self.pos[:, b]) < self.size[a] + self.size[b]

    @staticmethod
    def _bound(x):
        """simple_world_comm.py:169-174 / simple_tag.py:103-108."""
        with np.errstate(over="ignore"):
            far = np.minimum(np.exp(2 * x - 2), 10)
        return np.where(x < 0.9, 0, np.where(x < 1.0, (x - 0.9) * 10, far))

    # ---------------------------------------------------------------- Scenario.observation
    def observe(self):
        s = self.spec
        A, E, B = s.n_agents, s.n_entities, self.B
        name = s.name
        lm = lambda i: [self.pos[:, l] - self.pos[:, i] for l in range(A, E)]
        others = lambda i: [self.pos[:, j] - self.pos[:, i] for j in range(A) if j != i]
        out = []
        if name == "simple_adversary":          # :121-139
            goal = self._goal_pos(0)
            for i in range(A):
                cols = lm(i) + others(i)
                if not s.adversary[i]:
                    cols = [goal - self.pos[:, i]] + cols
                out.append(np.concatenate(cols, axis=1))
        elif name == "simple_push":             # :78-96; colours from reset_world :34-49
            goal = self._goal_pos(0)
            L = s.n_landmarks
            lm_color = []
            for l in range(L):
                col = np.full((B, 3), 0.1, self.dt_)
                col[:, l + 1] = self.dt_.type(0.1 + 0.8)     # color[i + 1] += 0.8 on a float64 array, then observed
                lm_color.append(col)
            for i in range(A):
                if s.adversary[i]:
                    cols = [self.vel[:, i]] + lm(i) + others(i)
                else:
                    mine = self._onehot(self.choice[:, 0] + 1, 3, self.dt_.type(0.25 + 0.5), 0.25)
                    cols = [self.vel[:, i], goal - self.pos[:, i], mine] + lm(i) + lm_color + others(i)
                out.append(np.concatenate(cols, axis=1))
        elif name == "simple_speaker_listener":  # :69-92; landmark colours :47-49
            out.append(self._onehot(self.choice[:, 0], 3, 0.65, 0.15))
            out.append(np.concatenate([self.vel[:, 1]] + lm(1) + [self.c[0]], axis=1))
        elif name == "simple_reference":        # :63-83; landmark colours :43-45
            for i in range(A):
                goal_b = self._onehot(self.choice[:, i], 3, 0.75, 0.25)
                out.append(np.concatenate([self.vel[:, i]] + lm(i) + [goal_b, self.c[1 - i]], axis=1))
        elif name == "simple_crypto":           # :127-169; colours are one-hots of width dim_c (:57-61)
            goal = self._onehot(self.choice[:, 0], s.dim_c, 1.0, 0.0)
            key = self._onehot(self.choice[:, 1], s.dim_c, 1.0, 0.0)
            said = self.c[2]                    # the speaker's utterance: the only `other.speaker`
            out = [said.copy(), np.concatenate([key, said], axis=1), np.concatenate([goal, key], axis=1)]
        elif name == "simple_world_comm":       # :231-289
            f1 = [self._hit(i, A + 3) for i in range(A)]
            f2 = [self._hit(i, A + 4) for i in range(A)]
            zero2 = np.zeros((B, 2), self.dt_)
            for i in range(A):
                leader = i == 0
                opos, ovel = [], []
                for j in range(A):
                    if j == i:
                        continue
                    vis = (f1[i] & f1[j]) | (f2[i] & f2[j]) | (~f1[i] & ~f1[j] & ~f2[i] & ~f2[j])
                    if leader:
                        vis = np.ones(B, bool)
                    opos.append(np.where(vis[:, None], self.pos[:, j] - self.pos[:, i], zero2))
                    if not s.adversary[j]:
                        ovel.append(np.where(vis[:, None], self.vel[:, j], zero2))
                inf = [np.where(f1[i], 1.0, -1.0).astype(self.dt_)[:, None], np.where(f2[i], 1.0, -1.0).astype(self.dt_)[:, None]]
                head = [self.vel[:, i], self.pos[:, i]] + lm(i) + opos
                if s.adversary[i]:
                    cols = head + ovel + inf + [self.c[0]]
                else:
                    cols = head + inf + ovel
                out.append(np.concatenate(cols, axis=1))
        else:
            raise KeyError(name)
        return out

    # ---------------------------------------------------------------- Scenario.reward (+ benchmark_data where it is data)
    def outputs(self):
        s = self.spec
        A, E, B = s.n_agents, s.n_entities, self.B
        name = s.name
        obs = self.observe()
        rew = np.zeros((A, B), self.dt_)
        info = {}
        advs = [j for j in range(A) if s.adversary[j]]
        good = [j for j in range(A) if not s.adversary[j]]
        if name == "simple_adversary":          # :76-119
            goal = self._goal_pos(0)
            adv_rew = 0
            for a in advs:                      # sum([...]) starts from int 0
                adv_rew = adv_rew + self._dist(self.pos[:, a], goal)
            gd = np.stack([self._dist(self.pos[:, a], goal) for a in good], axis=0)
            pos_rew = -gd.min(axis=0)
            for i in range(A):
                rew[i] = -self._d2(self.pos[:, i], goal) if s.adversary[i] else pos_rew + adv_rew
            # benchmark_data :57-67: adversary -> its squared goal distance; good -> squared distances to each landmark + goal
            info["adv_goal_d2"] = np.stack([self._d2(self.pos[:, a], goal) for a in advs], axis=0)
            info["good_d2"] = np.stack([np.stack([self._d2(self.pos[:, g], self.pos[:, l]) for l in range(A, E)] +
                                                 [self._d2(self.pos[:, g], goal)], axis=0) for g in good], axis=0)
        elif name == "simple_push":             # :60-76
            goal = self._goal_pos(0)
            gd = np.stack([self._dist(self.pos[:, a], goal) for a in good], axis=0)
            for i in range(A):
                if s.adversary[i]:
                    rew[i] = gd.min(axis=0) - self._dist(goal, self.pos[:, i])
                else:
                    rew[i] = -self._dist(self.pos[:, i], goal)
        elif name == "simple_speaker_listener":  # :63-67: the listener's squared distance to the goal landmark, for both
            r = -self._d2(self.pos[:, 1], self._goal_pos(0))
            rew[0] = r
            rew[1] = r
        elif name == "simple_reference":        # :57-61: agent i wants the OTHER agent at i's goal landmark
            for i in range(A):
                rew[i] = -self._d2(self.pos[:, 1 - i], self._goal_pos(i))
        elif name == "simple_crypto":           # :97-124
            goal = self._onehot(self.choice[:, 0], s.dim_c, 1.0, 0.0)

            def err(a):                         # np.sum(np.square(c - goal.color)); skipped when the utterance is all zeros
                e = np.sum(np.square(self.c[a] - goal), axis=1)
                return np.where((self.c[a] == 0).all(axis=1), 0.0, e)
            e_eve, e_bob = err(0), err(1)
            rew[0] = 0 - e_eve                  # adversary_reward :118-124
            good_rew = 0 - e_bob                # agent_reward :101-116: good_listeners = [Bob], adversaries = [Eve]
            adv_rew = 0 + e_eve
            rew[1] = adv_rew + good_rew
            rew[2] = adv_rew + good_rew
        elif name == "simple_world_comm":       # :143-203
            hit = {(g, v): self._hit(g, v) for g in good for v in advs}
            for i in range(A):
                r = np.zeros(B, self.dt_)
                if s.adversary[i]:              # adversary_reward :188-203 (shape = True)
                    dmin = np.stack([self._dist(self.pos[:, g], self.pos[:, i]) for g in good], axis=0).min(axis=0)
                    r = r - 0.1 * dmin
                    if s.collide[i]:
                        for g in good:
                            for v in advs:
                                r = r + np.where(hit[(g, v)], 5, 0)
                else:                           # agent_reward :156-186 (shape = False)
                    if s.collide[i]:
                        for v in advs:
                            r = r - np.where(hit[(i, v)], 5, 0)
                    for p in range(2):
                        r = r - 2 * self._bound(np.abs(self.pos[:, i, p]))
                    for f in (A + 1, A + 2):
                        r = r + np.where(self._hit(i, f), 2, 0)
                    fd = np.stack([self._dist(self.pos[:, f], self.pos[:, i]) for f in (A + 1, A + 2)], axis=0)
                    r = r + 0.05 * fd.min(axis=0)
                rew[i] = r
            coll = np.zeros((A, B), np.int32)   # benchmark_data :115-124
            for v in advs:
                for g in good:
                    coll[v] += hit[(g, v)]
            info["collisions"] = coll
            f1 = np.stack([self._hit(i, A + 3) for i in range(A)], axis=0)
            f2 = np.stack([self._hit(i, A + 4) for i in range(A)], axis=0)
            info["in_forest"] = np.stack([f1, f2], axis=0)   # (coverage statistics of the tests)
        else:
            raise KeyError(name)
        if s.collaborative:                     # environment.py:100-102
            total = np.sum(np.ascontiguousarray(rew.T), axis=1)
            rew = np.broadcast_to(total, (A, B)).copy()
        done = np.zeros((A, B), bool)
        return obs, rew, done, info


def branch_coverage(spec, g):
    """Share of a recorded trajectory's samples (golden .npz layout: pos [T,W,E,2], act<i>, c<i>, choice) in each discrete
    branch of the scenario's callbacks; keys are branch names, values fractions in [0, 1]."""
    name = spec.name
    A, E = spec.n_agents, spec.n_entities
    pos = np.asarray(g["pos"])
    T, W = pos.shape[:2]
    size = np.asarray(spec.size)
    dist = lambda a, b: np.sqrt(np.square(pos[:, :, a] - pos[:, :, b]).sum(-1))
    cov = {}
    if spec.choice_pops:
        ch = np.asarray(g["choice"])
        for k, n in enumerate(spec.choice_pops):
            for v in range(n):
                cov["pick%d=%d" % (k, v)] = float((ch[:, k] == v).mean())
    coll = [e for e in range(E) if spec.collide[e]]
    if len(coll) > 1:
        hit = np.zeros((T, W), bool)
        for ai, a in enumerate(coll):
            for b in coll[ai + 1:]:
                if spec.movable[a] or spec.movable[b]:
                    hit |= dist(a, b) < size[a] + size[b]
        cov["world-steps with a contact"] = float(hit.mean())
    if any(m is not None for m in spec.max_speed):
        vel = np.asarray(g["vel"])
        sp = np.sqrt(np.square(vel).sum(-1))
        ms = np.array([np.inf if m is None else m for m in spec.max_speed])
        cov["world-steps with an agent at its speed limit"] = float((sp >= ms * (1 - 1e-9)).any(-1).mean())
    for i in range(A):
        if not spec.silent_of(i):
            c = np.asarray(g["c%d" % i])
            cov["agent %d says a one-hot word" % i] = float(((c == 1).sum(-1) == 1).mean())
    if name == "simple_crypto":
        for i in (0, 1):
            c = np.asarray(g["c%d" % i])
            cov["agent %d's word == goal" % i] = float((np.argmax(c, -1) == np.asarray(g["choice"])[None, :, 0]).mean())
    if name == "simple_world_comm":
        advs = [j for j in range(A) if spec.adversary[j]]
        good = [j for j in range(A) if not spec.adversary[j]]
        f1 = np.stack([dist(i, A + 3) < size[i] + size[A + 3] for i in range(A)], 0)
        f2 = np.stack([dist(i, A + 4) < size[i] + size[A + 4] for i in range(A)], 0)
        cov["agents in a forest"] = float((f1 | f2).mean())
        cov["agents in both forests"] = float((f1 & f2).mean())
        hidden = []
        for i in range(1, A):              # the leader sees everybody
            for j in range(A):
                if j != i:
                    vis = (f1[i] & f1[j]) | (f2[i] & f2[j]) | (~f1[i] & ~f1[j] & ~f2[i] & ~f2[j])
                    hidden.append(~vis)
        cov["(observer, other) pairs hidden by a forest"] = float(np.mean(hidden))
        x = np.abs(pos[:, :, good])        # [T, W, G, 2]
        cov["prey coordinates in [0.9, 1)"] = float(((x >= 0.9) & (x < 1.0)).mean())
        cov["prey coordinates >= 1"] = float((x >= 1.0).mean())
        tag = np.stack([dist(g_, v) < size[g_] + size[v] for g_ in good for v in advs], 0)
        cov["world-steps with a prey caught"] = float(tag.any(0).mean())
        food = np.stack([dist(g_, f) < size[g_] + size[f] for g_ in good for f in (A + 1, A + 2)], 0)
        cov["world-steps with a prey on food"] = float(food.any(0).mean())
    return cov


def knife_edge(spec, pos, tol=1e-6):
    """[B] bool: worlds in which one of the scenario's strict `dist < size + size` tests (the ones that switch an output:
    forest membership, caught prey, prey on food) has |dist - threshold| < tol -- an fp32 implementation may legitimately
    decide those differently from fp64.  Physics contacts are continuous and need no mask."""
    pos = np.asarray(pos, np.float64)
    B = pos.shape[0]
    edge = np.zeros(B, bool)
    if spec.name != "simple_world_comm":
        return edge
    A = spec.n_agents
    size = np.asarray(spec.size)
    advs = [j for j in range(A) if spec.adversary[j]]
    good = [j for j in range(A) if not spec.adversary[j]]
    pairs = [(i, f) for i in range(A) for f in (A + 3, A + 4)] + [(g, v) for g in good for v in advs] + \
            [(g, f) for g in good for f in (A + 1, A + 2)]
    for a, b in pairs:
        d = np.sqrt(np.square(pos[:, a] - pos[:, b]).sum(-1))
        edge |= np.abs(d - (size[a] + size[b])) < tol
    return edge
