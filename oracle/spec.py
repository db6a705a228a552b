"""Scenario constants for the oracle (test infrastructure -- see oracle/__init__.py).

Every number below is read off the reference's `make_world` bodies; nothing is shared with the
product package so that a transcription error on either side shows up as a parity failure.

  world defaults      multiagent/core.py:83-99   dt .1, damping .25, contact_force 1e2, margin 1e-3
  entity defaults     multiagent/core.py:27-51   size .05, collide True, mass 1, no accel/max_speed
  simple              multiagent/scenarios/simple.py:6-22
  simple_spread       multiagent/scenarios/simple_spread.py:7-29   (3/3 hard-coded there; N here)
  simple_tag          multiagent/scenarios/simple_tag.py:7-36
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


def by_name(name, **kw):
    return {"simple": simple, "simple_spread": simple_spread, "simple_tag": simple_tag}[name](**kw)
