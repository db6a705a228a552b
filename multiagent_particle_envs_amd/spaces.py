"""Shape descriptors standing in for the gym spaces the reference builds in
MultiAgentEnv.__init__ (environment.py:38-69): only what callers read (.n, .shape, .low/.high,
.sample()).  `gym` itself is not a dependency."""
import numpy as np


class Discrete(object):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def sample(self):
        return int(np.random.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high = low, high
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.dtype = dtype

    def sample(self):
        lo = -1e3 if np.isinf(np.min(self.low)) else self.low
        hi = +1e3 if np.isinf(np.max(self.high)) else self.high
        return np.random.uniform(lo, hi, self.shape).astype(self.dtype)

    def __repr__(self):
        return "Box%s" % (self.shape,)


class MultiDiscrete(object):
    """[[lo, hi], ...] per sub-action (reference: multiagent/multi_discrete.py:9-44)."""

    def __init__(self, array_of_param_array):
        self.low = np.array([x[0] for x in array_of_param_array])
        self.high = np.array([x[1] for x in array_of_param_array])
        self.num_discrete_space = self.low.shape[0]
        self.shape = (self.num_discrete_space,)

    def sample(self):
        r = np.random.rand(self.num_discrete_space)
        return [int(x) for x in np.floor((self.high - self.low + 1.) * r + self.low)]

    def __repr__(self):
        return "MultiDiscrete" + str(self.num_discrete_space)


class Tuple(object):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)
