"""Scenario constants for the oracle (test infrastructure -- see oracle/__init__.py).

Every number below is read off the reference's `make_world` bodies; nothing is shared with the
product package so that a transcription error on either side shows up as a parity failure.

  world defaults      multiagent/core.py:83-99   dt .1, damping .25, contact_force 1e2, margin 1e-3
  entity defaults     multiagent/core.py:27-51   size .05, collide True, mass 1, no accel/max_speed
  simple              multiagent/scenarios/simple.py:6-22
  simple_spread       multiagent/scenarios/simple_spread.py:7-29   (3/3 hard-coded there; N here)
  simple_tag          multiagent/scenarios/simple_tag.py:7-36
  simple_adversary    multiagent/scenarios/simple_adversary.py:8-33
  simple_push         multiagent/scenarios/simple_push.py:6-32
  simple_speaker_listener  multiagent/scenarios/simple_speaker_listener.py:6-32
  simple_reference    multiagent/scenarios/simple_reference.py:6-24
  simple_crypto       multiagent/scenarios/simple_crypto.py:21-47
  simple_world_comm   multiagent/scenarios/simple_world_comm.py:7-58
"""
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class Spec:
    name: str
    n_agents: int
    n_landmarks: int
    dim_c: int
    size: List[float]                    # per entity, agents first (core.py:103-104)
    movable: List[bool]
    collide: List[bool]
    accel: List[Optional[float]]         # per agent; None -> sensitivity 5.0 (environment.py:178-181)
    max_speed: List[Optional[float]]     # per agent; None -> no clamp (core.py:164)
    adversary: List[bool] = field(default_factory=list)
    mass: Optional[List[float]] = None   # per entity: Entity.mass = initial_mass (core.py:47-51); None -> 1.0 everywhere
    collaborative: bool = False          # shared reward = sum over agents (environment.py:100-102)
    landmark_range: float = 1.0          # reset: uniform(-r, +r) for landmarks
    dt: float = 0.1
    damping: float = 0.25
    contact_force: float = 1e2
    contact_margin: float = 1e-3
    silent: Optional[List[bool]] = None  # per agent; None -> every agent silent (no communication action)
    choice_pops: List[int] = field(default_factory=list)   # np.random.choice draws of reset_world, in order: population sizes
    landmark_ranges: Optional[List[float]] = None          # reset: per-landmark uniform range (None -> landmark_range for all)

    def silent_of(self, i):
        return True if self.silent is None else self.silent[i]

    def mass_of(self, e):
        return 1.0 if self.mass is None else self.mass[e]

    @property
    def n_entities(self):
        return self.n_agents + self.n_landmarks

    def obs_dims(self):
        A, L = self.n_agents, self.n_landmarks
        if self.name == "simple":
            return [2 + 2 * L] * A
        if self.name == "simple_spread":
            return [4 + 2 * L + 2 * (A - 1) + self.dim_c * (A - 1)] * A
        if self.name == "simple_tag":
            out = []
            for i in range(A):
                n_good_others = sum(1 for j in range(A) if j != i and not self.adversary[j])
                out.append(4 + 2 * L + 2 * (A - 1) + 2 * n_good_others)
            return out
        if self.name == "simple_adversary":     # simple_adversary.py:121-139: good agents see the goal first
            return [2 * L + 2 * (A - 1) + (0 if self.adversary[i] else 2) for i in range(A)]
        if self.name == "simple_push":          # simple_push.py:78-96
            return [2 + 2 * L + 2 * (A - 1) if self.adversary[i] else 2 + 2 + 3 + 2 * L + 3 * L + 2 * (A - 1) for i in range(A)]
        if self.name == "simple_speaker_listener":   # :69-92
            return [3, 2 + 2 * L + self.dim_c]
        if self.name == "simple_reference":     # :63-83
            return [2 + 2 * L + 3 + self.dim_c] * A
        if self.name == "simple_crypto":        # :127-169: Eve hears, Bob key + hears, Alice goal + key
            return [self.dim_c, 2 * self.dim_c, 2 * self.dim_c]
        if self.name == "simple_world_comm":    # :231-289
            n_good = sum(1 for a in self.adversary if not a)
            out = []
            for i in range(A):
                if self.adversary[i]:
                    out.append(4 + 2 * L + 2 * (A - 1) + 2 * n_good + 2 + self.dim_c)
                else:
                    out.append(4 + 2 * L + 2 * (A - 1) + 2 + 2 * (n_good - 1))
            return out
        raise KeyError(self.name)


def simple():
    # one non-colliding silent agent, one non-colliding landmark; dim_c stays at World's 0
    return Spec("simple", 1, 1, 0,
                size=[0.05, 0.05], movable=[True, False], collide=[False, False],
                accel=[None], max_speed=[None], adversary=[False])


def simple_spread(n=3, n_landmarks=None):
    m = n if n_landmarks is None else n_landmarks
    return Spec("simple_spread", n, m, 2,
                size=[0.15] * n + [0.05] * m,
                movable=[True] * n + [False] * m,
                collide=[True] * n + [False] * m,
                accel=[None] * n, max_speed=[None] * n, adversary=[False] * n,
                collaborative=True)


def simple_tag(n_adversaries=3, n_good=1, n_landmarks=2):
    A = n_adversaries + n_good
    adv = [i < n_adversaries for i in range(A)]
    return Spec("simple_tag", A, n_landmarks, 2,
                size=[0.075 if a else 0.05 for a in adv] + [0.2] * n_landmarks,
                movable=[True] * A + [False] * n_landmarks,
                collide=[True] * (A + n_landmarks),
                accel=[3.0 if a else 4.0 for a in adv],
                max_speed=[1.0 if a else 1.3 for a in adv],
                adversary=adv, landmark_range=0.9)


def simple_adversary(n_agents=3, n_adversaries=1):
    # n_adversaries adversaries first, then the good agents (size .15, nobody collides), n_agents - 1 landmarks (size .08);
    # one goal landmark per world (reference make_world: 3 agents, 1 adversary)
    A, L = n_agents, n_agents - 1
    return Spec("simple_adversary", A, L, 2,
                size=[0.15] * A + [0.08] * L, movable=[True] * A + [False] * L, collide=[False] * (A + L),
                accel=[None] * A, max_speed=[None] * A, adversary=[i < n_adversaries for i in range(A)], choice_pops=[L])


def simple_push():
    # 1 adversary + 1 good agent, both colliding, default sizes; 2 non-colliding landmarks; one goal landmark per world
    return Spec("simple_push", 2, 2, 2,
                size=[0.05] * 4, movable=[True, True, False, False], collide=[True, True, False, False],
                accel=[None] * 2, max_speed=[None] * 2, adversary=[True, False], choice_pops=[2])


def simple_speaker_listener():
    # agent 0: the immovable speaker; agent 1: the silent listener; 3 landmarks (size .04); shared reward
    return Spec("simple_speaker_listener", 2, 3, 3,
                size=[0.075] * 2 + [0.04] * 3, movable=[False, True, False, False, False], collide=[False] * 5,
                accel=[None] * 2, max_speed=[None] * 2, adversary=[False] * 2, collaborative=True,
                silent=[False, True], choice_pops=[3])


def simple_reference():
    # 2 agents that move AND speak (10 words), 3 landmarks, nothing collides; shared reward; two goal picks
    return Spec("simple_reference", 2, 3, 10,
                size=[0.05] * 5, movable=[True, True, False, False, False], collide=[False] * 5,
                accel=[None] * 2, max_speed=[None] * 2, adversary=[False] * 2, collaborative=True,
                silent=[False, False], choice_pops=[3, 3])


def simple_crypto():
    # Eve (adversary), Bob, Alice (speaker): nobody moves, everybody speaks (4 words); goal and key picks among 2 landmarks
    return Spec("simple_crypto", 3, 2, 4,
                size=[0.05] * 5, movable=[False] * 5, collide=[False] * 5,
                accel=[None] * 3, max_speed=[None] * 3, adversary=[True, False, False],
                silent=[False] * 3, choice_pops=[2, 2])


def simple_world_comm(n_good=2, n_adversaries=4):
    # n_adversaries adversaries (agent 0 the speaking leader) + n_good good agents; landmarks = [obstacle] + 2 food + 2 forests
    # (reference make_world: 4 + 2)
    A = n_good + n_adversaries
    adv = [True] * n_adversaries + [False] * n_good
    return Spec("simple_world_comm", A, 5, 4,
                size=[0.075 if a else 0.045 for a in adv] + [0.2, 0.03, 0.03, 0.3, 0.3],
                movable=[True] * A + [False] * 5,
                collide=[True] * A + [True, False, False, False, False],
                accel=[3.0 if a else 4.0 for a in adv], max_speed=[1.0 if a else 1.3 for a in adv],
                adversary=adv, silent=[False] + [True] * (A - 1), landmark_range=0.9)


# Team sizes other than the reference's make_world that the tests cover (name, n_agents, n_adversaries): simple_adversary with 2..6
# agents and 1 or 2 adversaries, simple_world_comm with 1..3 good agents and 2..5 adversaries (the reference's own 3/1 and 6/4 are
# the f3_* goldens).  tests/golden/gen_golden_shapes.py records a golden per entry from the reference's callbacks.
TEAM_SIZE_VARIANTS = [("simple_adversary", a, v) for a in range(2, 7) for v in (1, 2) if v < a and (a, v) != (3, 1)] + \
                     [("simple_world_comm", g + v, v) for g in (1, 2, 3) for v in (2, 3, 4, 5) if (g, v) != (2, 4)]


def team_size_spec(name, n_agents, n_adversaries):
    if name == "simple_world_comm":
        return simple_world_comm(n_good=n_agents - n_adversaries, n_adversaries=n_adversaries)
    return simple_adversary(n_agents=n_agents, n_adversaries=n_adversaries)


def by_name(name, **kw):
    return {"simple": simple, "simple_spread": simple_spread, "simple_tag": simple_tag,
            "simple_adversary": simple_adversary, "simple_push": simple_push,
            "simple_speaker_listener": simple_speaker_listener, "simple_reference": simple_reference,
            "simple_crypto": simple_crypto, "simple_world_comm": simple_world_comm}[name](**kw)
