"""Minimal stand-in for the `gym` package (NOT a product dependency).

The reference (`/root/reference/multiagent/environment.py:1-3`, `multi_discrete.py:6-9`,
`multiagent/__init__.py:4`) imports `gym`, which is not installed and cannot be (no network).
`tests/golden/gen_golden.py` puts this directory on `sys.path` so that the *unmodified*
reference can be imported in the build container to record golden vectors.  It only
provides the names the reference touches at import/construct time; no behaviour on the
step path comes from here.
"""
from . import spaces  # noqa: F401


class Env(object):
    metadata = {}


class Space(object):
    def __init__(self, shape=None, dtype=None):
        self.shape = shape
        self.dtype = dtype
