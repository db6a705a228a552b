"""Scenario registry (reference: multiagent/scenarios/__init__.py:5-7 loads a scenario *file* by
name with imp.load_source; here `load("simple_spread.py")` resolves the built-in module of that
name, or executes a user file path, and returns the module -- callers do `.Scenario()` on it).

A user file may be written against either contract: this package's batched one (scenario.py) or the REFERENCE's own
(`from multiagent.core import World, Agent, Landmark`, `make_world(self)`, NumPy callbacks) -- for the latter the
`multiagent` import alias (compat/) is installed before the file is executed, and make_env wraps its Scenario in
refstyle.RefScenarioAdapter."""
import importlib
import importlib.util
import os.path as osp


def load(name):
    base = osp.basename(name)
    stem = base[:-3] if base.endswith(".py") else base
    here = osp.dirname(__file__)
    if osp.dirname(name) in ("", here) and osp.exists(osp.join(here, stem + ".py")):
        return importlib.import_module(__name__ + "." + stem)
    if osp.exists(name):  # a user-written scenario file
        from .. import compat
        compat.install()      # `import multiagent.core` / `multiagent.scenario` resolve (no-op when the reference itself is imported)
        spec = importlib.util.spec_from_file_location("mpe_user_scenario_" + stem, name)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    raise FileNotFoundError("no scenario %r (built-in: the reference's nine -- simple, simple_spread, simple_tag, "
                            "simple_adversary, simple_push, simple_speaker_listener, simple_reference, simple_crypto, "
                            "simple_world_comm; any other Scenario runs on the generic path)" % name)
