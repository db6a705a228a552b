"""Scenario plug-in protocol (reference: multiagent/scenario.py:4-10, README "Creating new
environments").  A scenario builds a World and supplies reset/reward/observation callbacks; with
`batch_size=B` every per-world scalar of the reference becomes a leading-B tensor."""


class BaseScenario(object):
    # create elements of the world
    def make_world(self, batch_size=1, device=None):
        raise NotImplementedError()

    # create initial conditions of the world
    def reset_world(self, world):
        raise NotImplementedError()
