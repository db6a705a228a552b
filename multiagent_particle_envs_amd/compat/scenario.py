"""`multiagent.scenario` stand-in: the base class reference-style Scenario files derive from (scenario.py:4-10)."""


class BaseScenario(object):
    """make_world(self) -> World and reset_world(self, world) are the two hooks a scenario must supply; reward /
    observation / benchmark_data / done are looked up by make_env."""

    def make_world(self):
        raise NotImplementedError("%s.make_world" % type(self).__name__)

    def reset_world(self, world):
        raise NotImplementedError("%s.reset_world" % type(self).__name__)
