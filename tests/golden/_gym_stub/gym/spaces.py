"""Shape-only stand-ins for gym.spaces used by the reference's MultiAgentEnv.__init__."""
import numpy as np


class _Prng(object):
    np_random = np.random.RandomState()


prng = _Prng()


class Discrete(object):
    def __init__(self, n):
        self.n = n
        self.shape = ()


class Box(object):
    def __init__(self, low, high, shape=None, dtype=None):
        self.low, self.high, self.shape, self.dtype = low, high, shape, dtype


class Tuple(object):
    def __init__(self, spaces):
        self.spaces = spaces
