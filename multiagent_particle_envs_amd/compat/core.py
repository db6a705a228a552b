"""`multiagent.core` stand-in: ONE world's entities as plain Python objects holding NumPy vectors -- the data model a
reference-style Scenario file is written against (core.py:4-99: attribute names and defaults are the contract).

These classes carry no physics.  A world made of them is a per-world VIEW of the batched device state: the adapter
(`refstyle.RefScenarioAdapter`) writes each world's post-step positions / velocities / utterances into them before it
calls the file's NumPy callbacks, and reads them back after `reset_world`.
"""


class _Attrs(object):
    """Instance attributes from a class-level table of defaults (mutable defaults are built per instance by `_fresh`)."""
    _defaults = {}

    def __init__(self):
        for klass in reversed(type(self).__mro__):
            self.__dict__.update(getattr(klass, "_defaults", {}))
        self._fresh()

    def _fresh(self):
        pass


class EntityState(_Attrs):
    _defaults = {"p_pos": None, "p_vel": None}


class AgentState(EntityState):
    _defaults = {"c": None}            # the utterance other agents observe


class Action(_Attrs):
    _defaults = {"u": None, "c": None}


class Entity(_Attrs):
    _defaults = {"name": "", "size": 0.050, "movable": False, "collide": True, "density": 25.0, "color": None,
                 "max_speed": None, "accel": None, "initial_mass": 1.0}

    def _fresh(self):
        self.state = EntityState()

    @property
    def mass(self):
        return self.initial_mass


class Landmark(Entity):
    pass


class Agent(Entity):
    _defaults = {"movable": True, "silent": False, "blind": False, "u_noise": None, "c_noise": None, "u_range": 1.0,
                 "action_callback": None}

    def _fresh(self):
        self.state = AgentState()
        self.action = Action()


class World(_Attrs):
    _defaults = {"dim_c": 0, "dim_p": 2, "dim_color": 3, "dt": 0.1, "damping": 0.25, "contact_force": 1e+2,
                 "contact_margin": 1e-3}

    def _fresh(self):
        self.agents = []
        self.landmarks = []

    @property
    def entities(self):
        return self.agents + self.landmarks

    @property
    def policy_agents(self):
        return [a for a in self.agents if a.action_callback is None]

    @property
    def scripted_agents(self):
        return [a for a in self.agents if a.action_callback is not None]

    def step(self):
        raise RuntimeError("this World is one world's NumPy view of a batched device state: it is stepped by "
                           "MultiAgentEnv.step (libmpe_hip.so, `mpe_world_step`); there is no CPU physics")
