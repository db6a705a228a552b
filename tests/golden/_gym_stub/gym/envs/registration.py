"""No-op stand-ins for gym.envs.registration (reference multiagent/__init__.py:4,9-21)."""


def register(*args, **kwargs):
    return None


class EnvSpec(object):
    def __init__(self, *args, **kwargs):
        pass
