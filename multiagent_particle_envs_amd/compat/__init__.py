"""`multiagent` import alias for reference-style Scenario files.

The reference's plug-in contract is a FILE (multiagent/scenarios/__init__.py:5-7, make_env.py:36-43, README "Creating new
environments") that does

    from multiagent.core import World, Agent, Landmark
    from multiagent.scenario import BaseScenario

and defines `Scenario.make_world(self)`, `reset_world(self, world)`, `reward(self, agent, world)`,
`observation(self, agent, world)` on ONE world of NumPy 2-vectors.  `install()` makes those two imports resolve to
`compat.core` / `compat.scenario` -- per-world data classes with the reference's attribute names (core.py:4-99,
scenario.py:4-10) and no physics: `World.step` belongs to the device (`refstyle.RefScenarioAdapter` steps the B worlds
through `mpe_world_step` and hands each world's NumPy views to the file's callbacks).

If a package called `multiagent` is already imported (the reference itself, installed by the user), it is left alone:
the adapter is duck-typed and works on the reference's own classes too.
"""
import sys
import types


def install(force=False):
    """Make `import multiagent.core` / `import multiagent.scenario` resolve to this package's stand-ins.  Returns True
    when the alias was installed by this call, False when some `multiagent` was already there."""
    if not force and "multiagent.core" in sys.modules and "multiagent.scenario" in sys.modules:
        return False
    from . import core, scenario
    pkg = sys.modules.get("multiagent")
    if pkg is None or force:
        pkg = types.ModuleType("multiagent")
        pkg.__doc__ = "alias installed by multiagent_particle_envs_amd.compat: reference-style Scenario files import from here"
        pkg.__path__ = []          # a package, with nothing to find on disk
        pkg.__mpe_alias__ = True
        sys.modules["multiagent"] = pkg
    sys.modules["multiagent.core"] = core
    sys.modules["multiagent.scenario"] = scenario
    pkg.core, pkg.scenario = core, scenario
    return True


def installed():
    return getattr(sys.modules.get("multiagent"), "__mpe_alias__", False)
