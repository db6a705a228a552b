"""Reference-style Scenario files, unmodified, at kernel speed: the file's NumPy callbacks TRACED into row programs.

A reference-style scenario (multiagent/scenario.py:4-10, README "Creating new environments") answers
`observation(agent, world)` / `reward(agent, world)` with NumPy arithmetic on ONE world's Python objects.  refstyle.py
runs such a file as it is -- B shadow worlds, the callbacks per world on the host: correct, and 3-5 orders of magnitude
slower than a kernel.  This module makes the same file fast without touching it:

  trace     the callbacks are called ONCE per agent (per control-flow path) on a shadow world whose state vectors hold
            symbolic scalars instead of floats (NumPy object arrays: `a - b`, np.square / sum / sqrt / exp, concatenate work
            unchanged); every arithmetic step becomes a node of an expression graph over the world's state -- positions,
            velocities, utterances, and the per-world PICKS `reset_world` draws with np.random.choice (a goal landmark, a
            key: objects chosen per world are proxies whose attributes are selections by the pick).  Control flow on the
            state (`if dist < dist_min: rew -= 1`): what only chooses between VALUES -- conditional expressions, `if`s that
            assign or append, early returns, `if T: continue`, and / or / not, comparisons of whole arrays and their reducing
            methods (`D < 0.2`, `D.min(axis=0)`, `.all()`) -- is predicated into select nodes by re-compiling a copy of the
            file's source (predicated_twin); any other symbolic `if` forks: the callback is re-run with the decision forced
            each way and the outcomes merge into select nodes, column by column (conditions are memoised per path, so a
            test asked twice forks once).  `reset_world` is traced the same way with np.random replaced by a recorder:
            initial positions as functions of uniform draws and picks; a reset_world that is no formula of its draws
            (rejection sampling, randn, shuffle) stays the file's own Python, run per restarting world at reset time
            (`Traced.host_reset`), its picks still followed; `benchmark_data` and `done` on request.
  verify    the graphs are evaluated with NumPy (fp64, vectorised over worlds) against the FILE'S OWN callbacks run
            concretely on random worlds -- states that touch, overlap, leave the arena -- before anything is generated:
            a callback that keeps hidden state, draws random numbers or does something the tracer does not model is
            caught here, and the env falls back to the host path of refstyle.py (always correct).
  generate  each agent's graphs become straight-line device code (`traced_obs` / `traced_rew`: one statement per node,
            in trace order = the reference's arithmetic order, fp32; `sqrt(x) < r` tests as the library's exact
            `sqrt_lt`), appended to the generated header of a compiled row program (rowspec.py, csrc/mpe_rows.hip:
            the ops MPE_ROW_OBS_CODE / MPE_ROW_R_CODE call it): `World.step`, the rows, the rewards -- ONE launch per
            step, the episode ends / rollouts of row programs included.

What is traced is arithmetic on the state: files whose callbacks read actions, scripted agents, movable landmarks, noise,
more than MPE_MAX_CHOICES picks, or more control-flow paths than `MAX_PATHS` stay on the host path.
Nothing here runs on the step path: tracing happens once, at env construction -- and while it does, np.random's drawing functions,
the math module's functions and the NumPy functions whose object-dtype loops cannot take symbolic values (maximum / minimum / clip /
min / max / where / any / all / sort / argmin / sqrt / exp / floor / sign ...: each is NumPy's own for ordinary numbers) are replaced
process-wide, and the traced file's own namespace sees min / max / any / all / float / int / round / sorted and numpy's array
constructors (np.zeros ...: through a proxy bound to its `np`; numpy itself keeps its own) that let symbolic values through --
all restored afterwards, the caller's random stream untouched: construct envs from one thread.
"""
import contextlib
import math
import struct
import sys

import numpy as np


class TraceUnsupported(Exception):
    """The file does something the tracer does not model: the caller falls back to the host path (refstyle.py)."""


MAX_PATHS = 8192          # control-flow paths of ONE callback of one agent (every symbolic `if` can double them)
MAX_DECISIONS = 4096      # symbolic decisions on ONE path
PARAM_POP = 1 << 24       # a uniform draw the callbacks read (a goal kept as coordinates, not as an entity): a "pick" among 2^24 values
MAX_SLOTS = 4             # per-world pick slots of the ABI (MPE_MAX_CHOICES): picks + such parameters
_CMP = ("lt", "le", "eq", "ne")
_BOOL_OPS = _CMP + ("not", "and", "or", "bconst")


def _np_mod(x, y):
    """x % y as NumPy answers it for floats (the sign of the divisor; nan for y == 0 where Python raises)."""
    with np.errstate(all="ignore"):
        return float(np.mod(np.float64(x), np.float64(y)))


class Node(object):
    """One value of the expression graph (hash-consed per Graph: structurally equal nodes are the same object)."""
    __slots__ = ("op", "args", "value", "uid", "is_bool")

    def __repr__(self):
        if self.op == "const":
            return repr(self.value)
        if self.op in ("P", "V", "C", "K", "U"):
            return "%s%s" % (self.op, list(self.value))
        return "%s(%s)" % (self.op, ", ".join(repr(a) for a in self.args))


class Graph(object):
    def __init__(self):
        self.nodes = {}
        self.count = 0

    def node(self, op, args=(), value=None):
        # (constants are keyed by bit pattern: -0.0 == 0.0 and they hash alike, but 1 / x and atan2 tell them apart)
        key = (op, tuple(a.uid for a in args), struct.pack("<d", value) if op == "const" else value)
        n = self.nodes.get(key)
        if n is None:
            n = Node()
            n.op, n.args, n.value, n.uid = op, tuple(args), value, self.count
            n.is_bool = op in _BOOL_OPS or (op == "ite" and args[1].is_bool)
            self.count += 1
            self.nodes[key] = n
        return n

    def const(self, v):
        if isinstance(v, (bool, np.bool_)):
            return self.node("bconst", (), bool(v))
        return self.node("const", (), float(v))

    # ---- constructors with constant folding (fp64, as the reference computes) ------------------------------------------------
    def unary(self, op, a):
        if a.op == "const":
            x = a.value
            try:
                v = {"neg": lambda: -x, "abs": lambda: abs(x), "sqrt": lambda: math.sqrt(x), "exp": lambda: math.exp(x),
                     "log": lambda: math.log(x), "tanh": lambda: math.tanh(x), "sin": lambda: math.sin(x), "cos": lambda: math.cos(x),
                     "floor": lambda: float(math.floor(x)), "rint": lambda: float(np.rint(x)), "f32": lambda: float(np.float32(x))}[op]()
            except (ValueError, OverflowError):
                v = float("nan")
            return self.const(v)
        return self.node(op, (a,))

    def binary(self, op, a, b):
        if a.op == "const" and b.op == "const":
            x, y = a.value, b.value
            try:
                v = {"add": lambda: x + y, "sub": lambda: x - y, "mul": lambda: x * y, "div": lambda: x / y,
                     "min": lambda: min(x, y), "max": lambda: max(x, y), "atan2": lambda: math.atan2(x, y),
                     "pow": lambda: math.pow(x, y), "mod": lambda: _np_mod(x, y)}[op]()
            except ZeroDivisionError:
                v = float("nan") if (x == 0 or op == "mod") else math.copysign(float("inf"), x)
            except (ValueError, OverflowError):
                v = float("nan")
            return self.const(v)
        if op == "sub" and a is b:         # an entity's offset to itself (simple_spread.py:78-81 counts the agent against itself)
            return self.const(0.0)
        return self.node(op, (a, b))

    def compare(self, op, a, b):
        if op == "gt":
            op, a, b = "lt", b, a
        elif op == "ge":
            op, a, b = "le", b, a
        if a.op == "const" and b.op == "const":
            x, y = a.value, b.value
            return self.const({"lt": x < y, "le": x <= y, "eq": x == y, "ne": x != y}[op])
        return self.node(op, (a, b))

    def lnot(self, a):
        if a.op == "bconst":
            return self.const(not a.value)
        if a.op == "not":
            return a.args[0]
        return self.node("not", (a,))

    def logical(self, op, a, b):
        if a.op == "bconst":
            return (b if a.value else a) if op == "and" else (a if a.value else b)
        if b.op == "bconst":
            return (a if b.value else b) if op == "and" else (b if b.value else a)
        if a is b:
            return a
        return self.node(op, (a, b))

    def ite(self, c, a, b):
        if a is b:
            return a
        if c.op == "bconst":
            return a if c.value else b
        if a.is_bool != b.is_bool:
            a, b = self.as_float(a), self.as_float(b)
        if a.is_bool and a.op == "bconst" and b.op == "bconst":     # ite(c, True, False) is c
            return c if a.value else self.lnot(c)
        return self.node("ite", (c, a, b))

    def as_float(self, a):
        if not a.is_bool:
            return a
        if a.op == "bconst":
            return self.const(1.0 if a.value else 0.0)
        return self.node("ite", (a, self.const(1.0), self.const(0.0)))

    def select(self, k, vals):
        """vals[pick k]: a per-world pick selects one of len(vals) values."""
        vals = list(vals)
        if all(v is vals[0] for v in vals):
            return vals[0]
        if any(v.is_bool for v in vals):
            vals = [self.as_float(v) for v in vals]
        return self.node("sel", (k,) + tuple(vals))


# ---- symbolic scalars: what a shadow world's state vectors hold while a callback is traced ----------------------------------
class _Ctx(object):
    graph = None
    tracer = None
    need_pick = None


def _lift(x):
    """anything a callback can combine with a Sym -> Node"""
    if isinstance(x, Sym):
        return x.n
    if isinstance(x, SymBool):
        return _Ctx.graph.as_float(x.n)
    if isinstance(x, (bool, np.bool_)):
        return _Ctx.graph.const(1.0 if x else 0.0)
    if isinstance(x, (int, float, np.integer, np.floating)):
        return _Ctx.graph.const(float(x))
    if isinstance(x, np.ndarray) and x.ndim == 0:
        return _lift(x.item())
    raise TraceUnsupported("a symbolic value combined with %r" % type(x).__name__)


def _select_nodes(c, a, b):
    return _Ctx.graph.ite(c, a, b)


class Sym(object):
    """A real number that depends on the world's state."""
    __slots__ = ("n",)

    def __init__(self, n):
        self.n = n

    def _b(self, op, o, swap=False):
        try:
            a, b = self.n, _lift(o)
        except TraceUnsupported:
            return NotImplemented
        return Sym(_Ctx.graph.binary(op, b, a) if swap else _Ctx.graph.binary(op, a, b))

    def __add__(self, o): return self._b("add", o)
    def __radd__(self, o): return self._b("add", o, True)
    def __sub__(self, o): return self._b("sub", o)
    def __rsub__(self, o): return self._b("sub", o, True)
    def __mul__(self, o): return self._b("mul", o)
    def __rmul__(self, o): return self._b("mul", o, True)
    def __truediv__(self, o): return self._b("div", o)
    def __rtruediv__(self, o): return self._b("div", o, True)
    def __neg__(self): return Sym(_Ctx.graph.unary("neg", self.n))
    def __pos__(self): return self
    def __abs__(self): return Sym(_Ctx.graph.unary("abs", self.n))

    def __pow__(self, o):
        if isinstance(o, (int, float, np.integer, np.floating)):
            if o == 2:
                return self * self
            if o == 1:
                return self
            if o == 0.5:
                return self.sqrt()
            if o == 3:
                return self * self * self
        return self._b("pow", o)          # (any other exponent, or a symbolic one: pow() as C computes it)

    def __rpow__(self, o): return self._b("pow", o, True)
    def __mod__(self, o): return self._b("mod", o)
    def __rmod__(self, o): return self._b("mod", o, True)

    # x // y as Python / NumPy compute it: (x - x % y) / y, an exact multiple of y up to rounding, to the nearest integer --
    # floor(x / y) is one too many where x / y rounds UP to an integer (1.0 // 0.1 is 9.0, floor(1.0 / 0.1) is 10.0)
    def __floordiv__(self, o):
        m = self._b("mod", o)
        return m if m is NotImplemented else ((self - m) / o).rint()

    def __rfloordiv__(self, o):
        m = self._b("mod", o, True)
        return m if m is NotImplemented else ((o - m) / self).rint()

    # rounding: floor / ceil / rint are what math.floor, math.ceil, np.floor, np.ceil, np.rint, np.round and round() ask for
    def floor(self): return Sym(_Ctx.graph.unary("floor", self.n))
    def ceil(self): return -((-self).floor())
    def rint(self): return Sym(_Ctx.graph.unary("rint", self.n))
    def trunc(self): return Sym(_select_nodes(_Ctx.graph.compare("lt", self.n, _Ctx.graph.const(0.0)), self.ceil().n, self.floor().n))
    __floor__, __ceil__, __trunc__ = floor, ceil, trunc

    def __round__(self, ndigits=None):      # (as np.round computes it: rint(x * 10^n) / 10^n -- round-half-even like Python's)
        if not ndigits:
            return self.rint()
        k = 10.0 ** int(ndigits)
        return (self * k).rint() / k

    def sign(self):
        g = _Ctx.graph
        zero = g.const(0.0)
        return Sym(_select_nodes(g.compare("lt", zero, self.n), g.const(1.0), _select_nodes(g.compare("lt", self.n, zero), g.const(-1.0), zero)))

    def f32(self): return Sym(_Ctx.graph.unary("f32", self.n))          # .astype(np.float32): rounded to single precision
    def isnan(self): return SymBool(_Ctx.graph.compare("ne", self.n, self.n))
    def isinf(self): return SymBool(_Ctx.graph.compare("eq", _Ctx.graph.unary("abs", self.n), _Ctx.graph.const(float("inf"))))
    def isfinite(self): return SymBool(_Ctx.graph.compare("lt", _Ctx.graph.unary("abs", self.n), _Ctx.graph.const(float("inf"))))
    def hypot(self, o): return (self * self + Sym(_lift(o)) * Sym(_lift(o))).sqrt()
    def square(self): return self * self
    def reciprocal(self): return 1.0 / self

    # the methods NumPy's object-dtype ufunc loops look for: np.sqrt(x) -> x.sqrt(), np.exp, np.log, np.tanh, np.square ...
    def sqrt(self): return Sym(_Ctx.graph.unary("sqrt", self.n))
    def exp(self): return Sym(_Ctx.graph.unary("exp", self.n))
    def log(self): return Sym(_Ctx.graph.unary("log", self.n))
    def tanh(self): return Sym(_Ctx.graph.unary("tanh", self.n))
    def sin(self): return Sym(_Ctx.graph.unary("sin", self.n))
    def cos(self): return Sym(_Ctx.graph.unary("cos", self.n))
    def arctan2(self, o): return self._b("atan2", o)
    def conjugate(self): return self

    def _c(self, op, o):
        try:
            return SymBool(_Ctx.graph.compare(op, self.n, _lift(o)))
        except TraceUnsupported:
            return NotImplemented

    def __lt__(self, o): return self._c("lt", o)
    def __le__(self, o): return self._c("le", o)
    def __gt__(self, o): return self._c("gt", o)
    def __ge__(self, o): return self._c("ge", o)
    def __eq__(self, o): return self._c("eq", o)
    def __ne__(self, o): return self._c("ne", o)
    __hash__ = object.__hash__

    def __bool__(self):          # truth of a number: x != 0
        return bool(SymBool(_Ctx.graph.compare("ne", self.n, _Ctx.graph.const(0.0))))

    def __float__(self):
        raise TraceUnsupported("a state-dependent value was converted to a Python float (float() / int() / formatting)")

    __int__ = __index__ = __float__

    def __repr__(self):
        return "Sym(%r)" % (self.n,)


class NeedConcretePick(Exception):
    """A per-world pick is used where Python needs a concrete value (an array index, a branch of reset_world): the trace is
    repeated once per value of that pick and the outcomes are merged by a selection on it."""

    def __init__(self, k):
        Exception.__init__(self, "pick %d" % k)
        self.k = k


class SymPickInt(Sym):
    """An integer attribute of a randomly chosen object (`goal.index`): a number like any Sym; used as an INDEX it cannot stay
    symbolic."""
    __slots__ = ("k",)

    def __init__(self, n, k):
        Sym.__init__(self, n)
        self.k = k

    def __index__(self):
        _Ctx.need_pick = self.k        # (NumPy turns whatever an __index__ raises into IndexError: the request travels beside it)
        raise NeedConcretePick(self.k)

    __int__ = __index__
    __hash__ = object.__hash__

    def _i(self, op, o, swap=False):
        r = Sym._b(self, op, o, swap)
        if r is not NotImplemented and isinstance(o, (int, np.integer)) and not isinstance(o, (bool, np.bool_)):
            return SymPickInt(r.n, self.k)       # goal.index + 1 is still an integer of that pick
        return r

    def __add__(self, o): return self._i("add", o)
    def __radd__(self, o): return self._i("add", o, True)
    def __sub__(self, o): return self._i("sub", o)
    def __rsub__(self, o): return self._i("sub", o, True)
    def __mul__(self, o): return self._i("mul", o)
    def __rmul__(self, o): return self._i("mul", o, True)


class SymBool(object):
    """A truth value that depends on the world's state.  Asking for its truth (`if`, `and`, `not`, bool()) FORKS the trace."""
    __slots__ = ("n",)

    def __init__(self, n):
        self.n = n

    def __bool__(self):
        if self.n.op == "bconst":
            return self.n.value
        if _Ctx.tracer is None:
            raise TraceUnsupported("a state-dependent condition outside a traced callback")
        return _Ctx.tracer.decide(self.n)

    def __invert__(self): return SymBool(_Ctx.graph.lnot(self.n))
    def __and__(self, o): return SymBool(_Ctx.graph.logical("and", self.n, _blift(o)))
    def __or__(self, o): return SymBool(_Ctx.graph.logical("or", self.n, _blift(o)))
    __rand__, __ror__ = __and__, __or__

    def __eq__(self, o):
        if isinstance(o, (bool, np.bool_)):
            return self if o else ~self
        return NotImplemented

    __hash__ = object.__hash__

    # arithmetic on a truth value (`hits += touching(a, b)`): 1.0 / 0.0
    def _f(self): return Sym(_Ctx.graph.as_float(self.n))
    def __add__(self, o): return self._f() + o
    def __radd__(self, o): return o + self._f()
    def __sub__(self, o): return self._f() - o
    def __rsub__(self, o): return o - self._f()
    def __mul__(self, o): return self._f() * o
    def __rmul__(self, o): return o * self._f()

    def __repr__(self):
        return "SymBool(%r)" % (self.n,)


def _blift(x):
    if isinstance(x, SymBool):
        return x.n
    if isinstance(x, (bool, np.bool_)):
        return _Ctx.graph.const(bool(x))
    raise TraceUnsupported("a symbolic condition combined with %r" % type(x).__name__)


def sym_min(*args, **kw):
    """`min` as the traced file sees it (injected into the module's globals): a MIN node instead of one fork per comparison."""
    return _minmax("min", min, args, kw)


def sym_max(*args, **kw):
    return _minmax("max", max, args, kw)


def first_extreme(xs, op):
    """The index of the first smallest (op "min") / largest element of `xs`, which hold symbolic values: ONE decision per
    candidate -- `x_k` beats everything before it strictly and everything after it weakly -- i.e. N paths for N elements, where
    NumPy's / Python's running comparison would fork 2^(N-1) ways."""
    xs = list(xs)
    for k in range(len(xs) - 1):
        if op == "min":
            c = sym_all([xs[k] < xs[j] for j in range(k)] + [xs[k] <= xs[j] for j in range(k + 1, len(xs))])
        else:
            c = sym_all([xs[k] > xs[j] for j in range(k)] + [xs[k] >= xs[j] for j in range(k + 1, len(xs))])
        if c:          # (forks)
            return k
    return len(xs) - 1


def sorted_values(xs):
    """`xs` (symbolic values among them) in ascending order WITHOUT a decision: a network of compare-exchanges (min, max) --
    N (N - 1) / 2 of them; the VALUES of a sort do not depend on how ties are broken."""
    xs = [x if isinstance(x, Sym) else Sym(_lift(x)) for x in xs]
    n = len(xs)
    for rnd in range(n):                              # odd-even transposition sort
        for k in range(rnd % 2, n - 1, 2):
            a, b = xs[k], xs[k + 1]
            xs[k], xs[k + 1] = sym_min(a, b), sym_max(a, b)
    return xs


def sym_sorted(it, key=None, reverse=False):
    items = list(it)
    if key is None and any(isinstance(x, (Sym, SymBool)) for x in items):
        out = sorted_values(items)
        return out[::-1] if reverse else out
    return sorted(items, key=key, reverse=reverse)          # (objects by a symbolic key: a genuine N!-way decision -- forks)


def _minmax(op, builtin, args, kw):
    if set(kw) == {"key"} and len(args) == 1:          # min(world.landmarks, key=lambda l: dist(agent, l)): the OBJECT whose key is smallest
        items = list(args[0])
        keys = [kw["key"](x) for x in items]
        if items and any(isinstance(k, (Sym, SymBool)) for k in keys):
            return items[first_extreme(keys, op)]
        return builtin(items, key=lambda x, _k=dict((id(x), k) for x, k in zip(items, keys)): _k[id(x)]) if items else builtin(items, **kw)
    if kw:
        return builtin(*args, **kw)
    items = list(args[0]) if len(args) == 1 else list(args)
    if not any(isinstance(x, (Sym, SymBool)) for x in items):
        return builtin(items)
    acc = _lift(items[0])
    for x in items[1:]:
        acc = _Ctx.graph.binary(op, acc, _lift(x))     # min(a, b) of Python: a if a <= b ... the same value for numbers
    return Sym(acc)


def _truthy(x):
    """the truth of one element as a node: a truth value itself, a number != 0"""
    if isinstance(x, SymBool):
        return x.n
    if isinstance(x, Sym):
        return _Ctx.graph.compare("ne", x.n, _Ctx.graph.const(0.0))
    return _Ctx.graph.const(bool(x))


def sym_any(it):
    items = list(it)
    if not any(isinstance(x, (Sym, SymBool)) for x in items):
        return any(items)
    acc = _Ctx.graph.const(False)
    for x in items:
        acc = _Ctx.graph.logical("or", acc, _truthy(x))
    return SymBool(acc)


def sym_all(it):
    items = list(it)
    if not any(isinstance(x, (Sym, SymBool)) for x in items):
        return all(items)
    acc = _Ctx.graph.const(True)
    for x in items:
        acc = _Ctx.graph.logical("and", acc, _truthy(x))
    return SymBool(acc)


class _NumberMeta(type):
    """isinstance(x, float) / issubclass(.., float) in the traced file keep answering as for the builtin type."""
    def __instancecheck__(cls, x):
        return isinstance(x, cls.__mro__[1])

    def __subclasscheck__(cls, k):
        return issubclass(k, cls.__mro__[1])


class sym_float(float, metaclass=_NumberMeta):
    """`float` as the traced file sees it: float(x) of a state-dependent number is that number, of a state-dependent truth value
    1.0 / 0.0 (`float(dist < r)`); anything else is the builtin's answer."""
    def __new__(cls, x=0.0):
        if isinstance(x, Sym):
            return x
        if isinstance(x, SymBool):
            return x._f()
        if isinstance(x, np.ndarray) and x.dtype == object and x.size == 1:
            return sym_float(x.reshape(-1)[0])
        return float(x)


class sym_int(int, metaclass=_NumberMeta):
    """`int`: int(truth value) -> 1.0 / 0.0 as a number of the graph, int(number) -> truncated towards zero."""
    def __new__(cls, x=0, *base):
        if isinstance(x, Sym) and not isinstance(x, SymPickInt):
            return x.trunc()
        if isinstance(x, SymBool):
            return x._f()
        if isinstance(x, np.ndarray) and x.dtype == object and x.size == 1 and not base:
            return sym_int(x.reshape(-1)[0])
        return int(x, *base)


def sym_round(x, ndigits=None):
    if isinstance(x, Sym):
        return x.__round__(ndigits)
    return round(x) if ndigits is None else round(x, ndigits)


_INJECTED = {"min": sym_min, "max": sym_max, "any": sym_any, "all": sym_all, "float": sym_float, "int": sym_int, "round": sym_round,
             "sorted": sym_sorted}


# ---- predication: `if`s that only assign, conditional expressions, early returns and and / or / not WITHOUT forking ---------------
# Forking doubles the paths with every symbolic `if`: a reward that tests every other agent (`if self.is_collision(a, agent):
# rew -= 1`, simple_spread.py:78-81) has 2^(N-1) of them.  Most such `if`s only choose between VALUES, and a value can be chosen
# without leaving the path: the file is re-compiled (its own source, a copy of its module; the file itself is not touched) with
#     X if T else Y                      ->  select(T, X, Y)                 when T is symbolic
#     if T: a = ..; b += ..  [else: ..]  ->  both branches run, a and b become select(T, then-value, else-value)
#     if T: return X   ... return Z      ->  return select(T, X, ... Z)
#     A and B, A or B, not A             ->  logical nodes
# whenever the test is symbolic, and exactly the original statement when it is an ordinary Python value; anything else (a branch that
# appends, continues, calls for side effects) still forks.  The twin is only TRACED; what the trace is verified against -- and what
# the host path runs -- is the original file's code.
_UNSET = object()


class _CannotMerge(Exception):
    pass


def _is_symbolic(x):
    return isinstance(x, (Sym, SymBool))


def _truth_node(t):
    if isinstance(t, SymBool):
        return t.n
    return _Ctx.graph.compare("ne", t.n, _Ctx.graph.const(0.0))


def _select_any(t, a, b):
    """the value `a if t else b` for a symbolic t, when a and b are values that can be merged"""
    g = _Ctx.graph
    if a is b:
        return a
    if a is _UNSET or b is _UNSET:
        # a name only ONE branch creates (`adv_l1 = ...` inside an else:, simple_crypto.py:112): a correct program reads it only
        # where that branch ran, so its value on the other side is nobody's business -- it keeps the one it has
        return b if a is _UNSET else a
    c = _truth_node(t)
    boolish = (bool, np.bool_, SymBool)
    if isinstance(a, boolish) and isinstance(b, boolish):
        return SymBool(g.ite(c, _blift(a), _blift(b)))
    scalar = (int, float, bool, np.integer, np.floating, np.bool_, Sym, SymBool)
    if isinstance(a, scalar) and isinstance(b, scalar):
        return Sym(g.ite(c, _lift(a), _lift(b)))
    if isinstance(a, np.ndarray) and isinstance(b, (np.ndarray, list, tuple)) or isinstance(b, np.ndarray) and isinstance(a, (list, tuple)):
        aa, bb = np.asarray(a, dtype=object), np.asarray(b, dtype=object)
        if aa.shape != bb.shape:
            raise _CannotMerge()
        out = np.empty(aa.size, dtype=object)
        out[:] = [_select_any(t, x, y) for x, y in zip(aa.reshape(-1), bb.reshape(-1))]
        return out.reshape(aa.shape)
    if isinstance(a, (list, tuple)) and type(a) is type(b) and len(a) == len(b):
        return type(a)(_select_any(t, x, y) for x, y in zip(a, b))
    if a is None or b is None or type(a) is not type(b):
        raise _CannotMerge()
    try:
        if a == b:
            return a
    except Exception:
        pass
    raise _CannotMerge()


def _p_ifexp(t, fa, fb):
    if not _is_symbolic(t):
        return fa() if t else fb()
    try:
        return _select_any(t, fa(), fb())
    except _CannotMerge:
        return fa() if bool(t) else fb()          # (forks)


def _p_and(*fs):
    acc = fs[0]()
    for f in fs[1:]:
        if not _is_symbolic(acc):
            if not acc:
                return acc
            acc = f()
            continue
        nxt = f()                                   # (no short circuit on a symbolic left side: the operands are expressions)
        if isinstance(nxt, (bool, np.bool_, SymBool)) or _is_symbolic(nxt):
            acc = SymBool(_Ctx.graph.logical("and", _truth_node(acc), _blift(nxt) if not isinstance(nxt, Sym) else _truth_node(nxt)))
        else:
            acc = nxt if bool(acc) else acc
    return acc


def _p_or(*fs):
    acc = fs[0]()
    for f in fs[1:]:
        if not _is_symbolic(acc):
            if acc:
                return acc
            acc = f()
            continue
        nxt = f()
        if isinstance(nxt, (bool, np.bool_, SymBool)) or _is_symbolic(nxt):
            acc = SymBool(_Ctx.graph.logical("or", _truth_node(acc), _blift(nxt) if not isinstance(nxt, Sym) else _truth_node(nxt)))
        else:
            acc = acc if bool(acc) else nxt
    return acc


def _p_not(x):
    if isinstance(x, SymBool):
        return ~x
    if isinstance(x, Sym):
        return SymBool(_Ctx.graph.lnot(_truth_node(x)))
    return not x


def _p_snap(x):
    """the value of a target as it is now (a list that a branch may append to: a copy)"""
    return list(x) if type(x) is list else x


def _p_cmp(op, a, b):
    """`a < b` (one operator) as the twin evaluates it: arrays that hold symbolic values are compared element by element into an
    array of truth values -- NumPy's own `<` on an object array stores bools, i.e. asks every element's comparison for its truth
    (one fork per element).  Anything else: the operator itself."""
    import operator
    f = getattr(operator, op)
    if ((isinstance(a, np.ndarray) and a.dtype == object) or (isinstance(b, np.ndarray) and b.dtype == object)) and _has_sym(a, b):
        mirrored = {"lt": "gt", "le": "ge", "gt": "lt", "ge": "le", "eq": "eq", "ne": "ne"}[op]

        def one(x, y):
            if isinstance(y, (Sym, SymBool)) and not isinstance(x, (Sym, SymBool)):
                return getattr(operator, mirrored)(y, x)
            return f(x, y)
        return _elementwise(one, a, b)
    return f(a, b)


def _p_method(name, obj, *args, **kw):
    """`x.min(axis=0)`, `x.any()`, `x.clip(lo, hi)`: the METHODS of an array that holds symbolic values answered by the functions
    np.min / np.any / np.clip as they are while a file is traced (_numpy_patches); any other object: its own method."""
    if name == "sort":          # in place: a list of numbers (`dists.sort()`), an array along its last axis
        if type(obj) is list and not args and set(kw) <= {"reverse"} and _has_sym(obj):
            obj[:] = sym_sorted(obj, reverse=kw.get("reverse", False))
            return None
        if isinstance(obj, np.ndarray) and obj.dtype == object and not args and not kw and _has_sym(obj):
            obj[...] = np.sort(obj)
            return None
        return obj.sort(*args, **kw)
    if name == "astype":        # obs.astype(np.float32): NumPy would ask every element for its __float__
        if isinstance(obj, np.ndarray) and obj.dtype == object and _has_sym(obj) and len(args) == 1 and not kw:
            return _as_dtype(obj, args[0])
        return obj.astype(*args, **kw)
    if isinstance(obj, np.ndarray) and obj.dtype == object and _has_sym(obj):
        return getattr(np, name)(obj, *args, **kw)
    return getattr(obj, name)(*args, **kw)


def _as_dtype(arr, dtype):
    """an array that holds symbolic values, "converted": to float32 -> rounded to single precision (a node); to float64 / float /
    object -> itself; to an integer type -> truncated"""
    if dtype in (sym_float, float, object) or dtype is np.float64:
        return arr.copy()
    if dtype is sym_int:
        dtype = int
    try:
        dt = np.dtype(dtype)
    except TypeError:
        raise TraceUnsupported("astype(%r) of state-dependent values" % (dtype,))
    if dt == np.float64 or dt == object:
        return arr.copy()
    if dt == np.float32:
        return _elementwise(lambda x: Sym(_lift(x)).f32(), arr)
    if dt.kind in "iu":
        return _elementwise(lambda x: Sym(_lift(x)).trunc() if isinstance(x, (Sym, SymBool)) else int(x), arr)
    raise TraceUnsupported("astype(%s) of state-dependent values" % dt)


_PREDICATION_HELPERS = {"_mpe_cmp": _p_cmp, "_mpe_method": _p_method, "_mpe_snap": _p_snap, "_mpe_sym": _is_symbolic, "_mpe_sel": _select_any, "_mpe_ifexp": _p_ifexp, "_mpe_and": _p_and,
                        "_mpe_or": _p_or, "_mpe_not": _p_not, "_MPE_UNSET": _UNSET, "_mpe_CannotMerge": _CannotMerge}


def _predicate_tree(tree):
    import ast

    counter = [0]

    def lam(expr):
        return ast.Lambda(args=ast.arguments(posonlyargs=[], args=[], vararg=None, kwonlyargs=[], kw_defaults=[], kwarg=None, defaults=[]),
                          body=expr)

    def call(name, *args):
        return ast.Call(func=ast.Name(id=name, ctx=ast.Load()), args=list(args), keywords=[])

    def simple_target(t):
        if isinstance(t, ast.Name):
            return True
        if isinstance(t, ast.Subscript):       # in_forest[0] = ...: an element of something that exists
            return isinstance(t.value, ast.Name) and isinstance(t.slice, (ast.Constant, ast.Name))
        return False

    def is_append(st):
        return isinstance(st, ast.Expr) and isinstance(st.value, ast.Call) and isinstance(st.value.func, ast.Attribute) and \
            st.value.func.attr == "append" and isinstance(st.value.func.value, ast.Name) and len(st.value.args) == 1 and \
            not st.value.keywords

    def assign_only(stmts):
        """statements that only assign to local names / constant-indexed elements, or append to local lists (ifs of such,
        recursively)"""
        for st in stmts:
            if isinstance(st, ast.Pass) or is_append(st):
                continue
            if isinstance(st, ast.Assign) and len(st.targets) == 1 and simple_target(st.targets[0]):
                continue
            if isinstance(st, ast.AugAssign) and simple_target(st.target):
                continue
            if isinstance(st, ast.If) and assign_only(st.body) and assign_only(st.orelse):
                continue
            return False
        return True

    def targets_of(stmts, out):
        import copy
        for st in stmts:
            if isinstance(st, ast.Assign):
                t = st.targets[0]
            elif isinstance(st, ast.AugAssign):
                t = st.target
            elif isinstance(st, ast.If):
                targets_of(st.body, out)
                targets_of(st.orelse, out)
                continue
            elif is_append(st):
                t = ast.Name(id=st.value.func.value.id, ctx=ast.Store())      # the list itself is the value that changes
            else:
                continue
            key = ast.dump(t)
            if key not in out:
                out[key] = copy.deepcopy(t)
        return out

    def as_load(t):
        import copy
        t = copy.deepcopy(t)
        for n in ast.walk(t):
            if hasattr(n, "ctx"):
                n.ctx = ast.Load()
        return t

    def as_store(t):
        import copy
        t = copy.deepcopy(t)
        t.ctx = ast.Store()
        return t

    class T(ast.NodeTransformer):
        def visit_IfExp(self, node):
            self.generic_visit(node)
            return call("_mpe_ifexp", node.test, lam(node.body), lam(node.orelse))

        def visit_BoolOp(self, node):
            self.generic_visit(node)
            return call("_mpe_and" if isinstance(node.op, ast.And) else "_mpe_or", *[lam(v) for v in node.values])

        def visit_UnaryOp(self, node):
            self.generic_visit(node)
            if isinstance(node.op, ast.Not):
                return call("_mpe_not", node.operand)
            return node

        def visit_Compare(self, node):
            self.generic_visit(node)
            names = {ast.Lt: "lt", ast.LtE: "le", ast.Gt: "gt", ast.GtE: "ge", ast.Eq: "eq", ast.NotEq: "ne"}
            if len(node.ops) == 1 and type(node.ops[0]) in names:
                return call("_mpe_cmp", ast.Constant(value=names[type(node.ops[0])]), node.left, node.comparators[0])
            if len(node.ops) > 1 and all(type(o) in names for o in node.ops) and \
                    not any(isinstance(n, (ast.Call, ast.NamedExpr, ast.Await, ast.Yield)) for m in node.comparators[:-1] for n in ast.walk(m)):
                import copy          # a < x < b with a plain middle operand (evaluating it twice changes nothing): a < x and x < b
                terms = [node.left] + list(node.comparators)
                return call("_mpe_and", *[lam(call("_mpe_cmp", ast.Constant(value=names[type(o)]), copy.deepcopy(terms[k]), copy.deepcopy(terms[k + 1])))
                                          for k, o in enumerate(node.ops)])
            return node          # (`is`, `in`, chains through calls: as written)

        def _filtered(self, node):
            """sum / len / min / max / any / all over ONE comprehension with `if` filters: the filter moved into the element
            (`sum(E for x in xs if C)` -> `sum(E if C else 0 ...)`), so that a symbolic C selects instead of deciding how long a list is.
            The same value whenever C is an ordinary truth value (min / max of nothing: +-inf instead of ValueError)."""
            if not (isinstance(node.func, ast.Name) and node.func.id in ("sum", "len", "min", "max", "any", "all") and len(node.args) == 1
                    and not node.keywords and isinstance(node.args[0], (ast.GeneratorExp, ast.ListComp))):
                return None
            comp = node.args[0]
            if len(comp.generators) != 1 or not comp.generators[0].ifs or comp.generators[0].is_async:
                return None
            gen = comp.generators[0]
            cond = gen.ifs[0] if len(gen.ifs) == 1 else ast.BoolOp(op=ast.And(), values=list(gen.ifs))
            kind = node.func.id
            inf = ast.Call(func=ast.Name(id="float", ctx=ast.Load()), args=[ast.Constant(value="inf")], keywords=[])
            if kind in ("sum", "len"):
                elt = ast.IfExp(test=cond, body=comp.elt if kind == "sum" else ast.Constant(value=1), orelse=ast.Constant(value=0))
                func = "sum"
            elif kind in ("min", "max"):
                elt = ast.IfExp(test=cond, body=comp.elt, orelse=inf if kind == "min" else ast.UnaryOp(op=ast.USub(), operand=inf))
                func = kind
            elif kind == "any":
                elt, func = ast.BoolOp(op=ast.And(), values=[cond, comp.elt]), "any"
            else:
                elt, func = ast.BoolOp(op=ast.Or(), values=[ast.UnaryOp(op=ast.Not(), operand=cond), comp.elt]), "all"
            new_gen = ast.comprehension(target=gen.target, iter=gen.iter, ifs=[], is_async=0)
            return ast.Call(func=ast.Name(id=func, ctx=ast.Load()), args=[ast.ListComp(elt=elt, generators=[new_gen])], keywords=[])

        def visit_Call(self, node):
            moved = self._filtered(node)
            if moved is not None:
                node = moved
            self.generic_visit(node)
            if isinstance(node.func, ast.Attribute) and node.func.attr in ("min", "max", "any", "all", "clip", "argmin", "argmax", "sort", "astype") and \
                    not any(isinstance(a, ast.Starred) for a in node.args) and not any(k.arg is None for k in node.keywords):
                return ast.Call(func=ast.Name(id="_mpe_method", ctx=ast.Load()),
                                args=[ast.Constant(value=node.func.attr), node.func.value] + list(node.args), keywords=list(node.keywords))
            return node

        def _returns_chain(self, stmts):
            """A block that consists of nothing but tests and returns -- `if T: return X` ... `return Z`, also nested (`if T: if U:
            return X; return Y` / elif / else) -- as the expression select(T, X, ... Z); None for anything else."""
            if len(stmts) == 1 and isinstance(stmts[0], ast.Return) and stmts[0].value is not None:
                return stmts[0].value
            st = stmts[0] if stmts else None
            if isinstance(st, ast.If):
                then = self._returns_chain(st.body)
                if then is not None:
                    rest = self._returns_chain(list(st.orelse) + list(stmts[1:]))
                    if rest is not None:
                        return call("_mpe_ifexp", st.test, lam(then), lam(rest))
            return None

        def _block(self, stmts):
            """a function body: early-return chains at its end become one return; `if T: return X` followed by more statements
            that end in the function's only other return becomes `if T: r = X / else: <the statements>, r = Z` + `return r` (which
            visit_If then predicates where the statements only assign)"""
            out = []
            for k, st in enumerate(stmts):
                if isinstance(st, ast.If):
                    chain = self._returns_chain(stmts[k:])
                    if chain is not None:
                        out.append(ast.Return(value=chain))
                        return out
                    if not st.orelse and len(st.body) == 1 and isinstance(st.body[0], ast.Return) and st.body[0].value is not None:
                        rest = self._block(list(stmts[k + 1:]))
                        if rest and isinstance(rest[-1], ast.Return) and rest[-1].value is not None and not self._returns_inside(rest[:-1]):
                            rv = "_mpe_ret%d" % counter[0]
                            counter[0] += 1

                            def setr(value):
                                return ast.Assign(targets=[ast.Name(id=rv, ctx=ast.Store())], value=value)
                            out.append(ast.If(test=st.test, body=[setr(st.body[0].value)], orelse=rest[:-1] + [setr(rest[-1].value)]))
                            out.append(ast.Return(value=ast.Name(id=rv, ctx=ast.Load())))
                            return out
                out.append(st)
            return out

        def _returns_inside(self, stmts):
            todo = list(stmts)
            while todo:
                n = todo.pop()
                if isinstance(n, (ast.Return, ast.Yield, ast.YieldFrom)):
                    return True
                if isinstance(n, (ast.FunctionDef, ast.Lambda, ast.ClassDef)):
                    continue
                todo.extend(ast.iter_child_nodes(n))
            return False

        def _decontinue(self, stmts):
            """`if T: continue [else: E]` followed by the rest of a loop body  ->  `if not T: [E] <rest>` (the same program; the rest
            can then be predicated where it only assigns: simple_crypto.py:104-113)"""
            for k, st in enumerate(stmts):
                if isinstance(st, ast.If) and len(st.body) == 1 and isinstance(st.body[0], ast.Continue):
                    rest = self._decontinue(list(st.orelse) + list(stmts[k + 1:]))          # (`else:` of a continue: the rest, too)
                    if not rest:
                        return list(stmts[:k]) + [ast.Expr(value=st.test)]
                    return list(stmts[:k]) + [ast.If(test=ast.UnaryOp(op=ast.Not(), operand=st.test), body=rest, orelse=[])]
            return list(stmts)

        def visit_For(self, node):
            node.body = self._decontinue(node.body)
            self.generic_visit(node)
            return node

        def visit_While(self, node):
            node.body = self._decontinue(node.body)
            self.generic_visit(node)
            return node

        def visit_FunctionDef(self, node):
            node.body = self._block(node.body)          # (on the file's own statements: what it builds is then visited like the rest)
            self.generic_visit(node)
            return node

        def visit_If(self, node):
            ok = assign_only(node.body) and assign_only(node.orelse)        # (judged on the file's own statements ...)
            tg = list(targets_of(node.body + node.orelse, {}).values()) if ok else []
            self.generic_visit(node)                                        # (... the nested ifs inside are predicated first)
            if not ok:
                return node
            k = counter[0]
            counter[0] += 1
            tv = "_mpe_t%d" % k
            stmts = [ast.Assign(targets=[ast.Name(id=tv, ctx=ast.Store())], value=node.test)]
            plain = ast.If(test=ast.Name(id=tv, ctx=ast.Load()), body=node.body, orelse=node.orelse)
            sym = []
            # snapshot (a name that does not exist yet: UNSET)
            for j, t in enumerate(tg):
                sv = "_mpe_s%d_%d" % (k, j)
                sym.append(ast.Try(body=[ast.Assign(targets=[ast.Name(id=sv, ctx=ast.Store())], value=call("_mpe_snap", as_load(t)))],
                                   handlers=[ast.ExceptHandler(type=ast.Tuple(elts=[ast.Name(id="NameError", ctx=ast.Load()),
                                                                                     ast.Name(id="IndexError", ctx=ast.Load()),
                                                                                     ast.Name(id="KeyError", ctx=ast.Load())], ctx=ast.Load()),
                                                               name=None,
                                                               body=[ast.Assign(targets=[ast.Name(id=sv, ctx=ast.Store())],
                                                                                value=ast.Name(id="_MPE_UNSET", ctx=ast.Load()))])],
                                   orelse=[], finalbody=[]))
            import copy

            def grab(prefix):
                res = []
                for j, t in enumerate(tg):
                    res.append(ast.Try(body=[ast.Assign(targets=[ast.Name(id="%s%d_%d" % (prefix, k, j), ctx=ast.Store())], value=call("_mpe_snap", as_load(t)))],
                                       handlers=[ast.ExceptHandler(type=ast.Name(id="NameError", ctx=ast.Load()), name=None,
                                                                   body=[ast.Assign(targets=[ast.Name(id="%s%d_%d" % (prefix, k, j), ctx=ast.Store())],
                                                                                    value=ast.Name(id="_MPE_UNSET", ctx=ast.Load()))])],
                                       orelse=[], finalbody=[]))
                return res

            def restore():
                res = []
                for j, t in enumerate(tg):
                    sv = "_mpe_s%d_%d" % (k, j)
                    # (a name that did not exist is left as the branch left it: it is UNSET in the snapshot and cannot merge unless both
                    #  branches define it)
                    gone = [ast.Pass()]
                    if isinstance(t, ast.Name):      # a name the branch created: it does not exist on the other side
                        gone = [ast.Try(body=[ast.Delete(targets=[ast.Name(id=t.id, ctx=ast.Del())])],
                                        handlers=[ast.ExceptHandler(type=ast.Name(id="NameError", ctx=ast.Load()), name=None, body=[ast.Pass()])],
                                        orelse=[], finalbody=[])]
                    res.append(ast.If(test=ast.Compare(left=ast.Name(id=sv, ctx=ast.Load()), ops=[ast.IsNot()],
                                                       comparators=[ast.Name(id="_MPE_UNSET", ctx=ast.Load())]),
                                      body=[ast.Assign(targets=[as_store(t)], value=call("_mpe_snap", ast.Name(id=sv, ctx=ast.Load())))],
                                      orelse=gone))
                return res
            merged = []
            for j, t in enumerate(tg):
                merged.append(ast.Assign(targets=[as_store(t)],
                                         value=call("_mpe_sel", ast.Name(id=tv, ctx=ast.Load()), ast.Name(id="_mpe_a%d_%d" % (k, j), ctx=ast.Load()),
                                                    ast.Name(id="_mpe_b%d_%d" % (k, j), ctx=ast.Load()))))
            sym += copy.deepcopy(node.body) + grab("_mpe_a") + restore() + (copy.deepcopy(node.orelse) or [ast.Pass()]) + grab("_mpe_b")
            # merge -- or, where a value cannot be merged (defined on one side only, objects): put the state back and fork
            sym.append(ast.Try(body=merged,
                               handlers=[ast.ExceptHandler(type=ast.Name(id="_mpe_CannotMerge", ctx=ast.Load()), name=None,
                                                           body=restore() + [copy.deepcopy(plain)])],
                               orelse=[], finalbody=[]))
            stmts.append(ast.If(test=call("_mpe_sym", ast.Name(id=tv, ctx=ast.Load())), body=sym, orelse=[plain]))
            return stmts

    tree = T().visit(tree)
    ast.fix_missing_locations(tree)
    return tree


def source_file(klass):
    """The file a class was defined in, through its methods' code objects (a scenario file executed by path is in no sys.modules,
    so inspect.getsourcefile does not find it)."""
    import inspect
    import types
    for f in klass.__dict__.values():
        f = getattr(f, "__func__", f)
        if isinstance(f, types.FunctionType):
            return f.__code__.co_filename
    return inspect.getsourcefile(klass)


def _twin_neighbours(space, path):
    """The user's own helper modules next to the file (`import helpers`, `from helpers import dist`), predicated the same way:
    the twin's namespace gets twins of those modules / of the functions it imported from them (one level; a helper that cannot
    be re-compiled stays as it is -- its `if`s then fork)."""
    import ast
    import os
    import types
    home = os.path.dirname(os.path.abspath(path))
    pkg = __name__.rsplit(".", 1)[0]
    twins = {}

    def twin_of(g):
        f = g.get("__file__")
        if not (isinstance(f, str) and os.path.abspath(f).startswith(home + os.sep)) or g.get("__name__", "").startswith(pkg + ".") \
                or os.path.abspath(f) == os.path.abspath(path):
            return None
        if id(g) not in twins:
            try:
                with open(f) as fh:
                    tree = _predicate_tree(ast.parse(fh.read(), filename=f))
                m = types.ModuleType(g.get("__name__", "helper") + "__predicated")
                m.__dict__.update({"__file__": f})
                m.__dict__.update(_PREDICATION_HELPERS)
                exec(compile(tree, f, "exec"), m.__dict__)
                twins[id(g)] = m
            except Exception:
                twins[id(g)] = None
        return twins[id(g)]
    for name, v in list(space.items()):
        if isinstance(v, types.ModuleType):
            m = twin_of(v.__dict__)
            if m is not None:
                space[name] = m
        elif isinstance(v, types.FunctionType) and v.__globals__ is not space:
            m = twin_of(v.__globals__)
            if m is not None and isinstance(getattr(m, v.__name__, None), types.FunctionType):
                space[name] = getattr(m, v.__name__)


def predicated_twin(scenario):
    """A twin of `scenario` whose class was re-compiled from its own source with value-only control flow predicated (above); the
    twin shares the scenario's instance attributes.  Raises TraceUnsupported when the source is not available / does not recompile."""
    import ast
    import inspect
    import sys
    klass = type(scenario)
    try:
        path = source_file(klass)
        with open(path) as fh:
            source = fh.read()
        tree = _predicate_tree(ast.parse(source, filename=path))
        code = compile(tree, path, "exec")
        space = {"__name__": klass.__module__ + "__predicated", "__file__": path}
        space.update(_PREDICATION_HELPERS)
        with patched_math():
            exec(code, space)
            _twin_neighbours(space, path)
        twin_class = space[klass.__name__]
    except TraceUnsupported:
        raise
    except Exception as e:
        raise TraceUnsupported("the file's source could not be re-compiled for predication (%s: %s)" % (type(e).__name__, e))
    twin = twin_class.__new__(twin_class)
    twin.__dict__.update(scenario.__dict__)
    return twin


# ---- control flow: every path of a callback, merged -------------------------------------------------------------------------
class Tracer(object):
    """Runs a callback once per control-flow path.  `decide` answers a symbolic condition: by the forced prefix while it
    lasts, then True; the conditions met and the answers given are the path.  `explore` flips the answers depth-first and
    merges the outcomes into select nodes."""

    def __init__(self, graph, max_paths=None, max_decisions=None):
        self.g = graph
        self.max_paths = MAX_PATHS if max_paths is None else max_paths
        self.max_decisions = MAX_DECISIONS if max_decisions is None else max_decisions
        self.paths = 0

    def decide(self, n):
        neg = False
        if n.op == "not":
            n, neg = n.args[0], True
        d = self.memo.get(n)
        if d is None:
            if len(self.decs) >= self.max_decisions:      # `while <state-dependent test>:` never ends when every new test is answered True
                raise TraceUnsupported("more than %d state-dependent decisions on one path (a loop that runs while a state-dependent "
                                       "condition holds?)" % self.max_decisions)
            d = self.prefix[len(self.decs)] if len(self.decs) < len(self.prefix) else True
            self.conds.append(n)
            self.decs.append(d)
            self.memo[n] = d
        return (not d) if neg else d

    def _run(self, fn, prefix):
        self.paths += 1
        if self.paths > self.max_paths:
            raise TraceUnsupported("more than %d control-flow paths in one callback (a symbolic `if` per entity in a large team?)"
                                   % self.max_paths)
        self.prefix, self.conds, self.decs, self.memo = prefix, [], [], {}
        keep, _Ctx.tracer = _Ctx.tracer, self
        try:
            out = fn()
        finally:
            _Ctx.tracer = keep
        if self.decs[:len(prefix)] != prefix:
            raise TraceUnsupported("the callback is not deterministic (a re-run took another path)")
        return out, self.conds, self.decs

    def explore(self, fn, normalise):
        """-> the merged outcome: `normalise(raw)` turns one path's return value into a list of Nodes (same length on every
        path)."""
        def build(prefix):
            raw, conds, decs = self._run(fn, prefix)
            res = normalise(raw)
            for d in range(len(conds) - 1, len(prefix) - 1, -1):
                other = build(decs[:d] + [False])
                if len(other) != len(res):
                    raise TraceUnsupported("the callback returns %d values on one path and %d on another" % (len(res), len(other)))
                res = [self.g.ite(conds[d], a, b) for a, b in zip(res, other)]
            return res
        return build([])


# ---- np.random while reset_world is traced (or replayed concretely) ---------------------------------------------------------
class PickProxy(object):
    """What `np.random.choice(objects)` returns while reset_world is traced: every attribute is the selection, by the pick, of
    that attribute over the population; an assignment lands on whichever object the pick names."""

    def __init__(self, k_node, population):
        object.__setattr__(self, "_k", k_node)
        object.__setattr__(self, "_pop", list(population))

    def __getattr__(self, name):
        k, pop = object.__getattribute__(self, "_k"), object.__getattribute__(self, "_pop")
        try:
            vals = [getattr(o, name) for o in pop]
        except AttributeError:
            raise TraceUnsupported("attribute %r of a randomly chosen object: not every candidate has it" % name)
        return _select_values(k, vals, name)

    def __setattr__(self, name, value):
        k, pop = object.__getattribute__(self, "_k"), object.__getattribute__(self, "_pop")
        g = _Ctx.graph
        for i, o in enumerate(pop):
            if not hasattr(o, name) or getattr(o, name) is None:
                raise TraceUnsupported("assignment to attribute %r of a randomly chosen object whose candidates do not all have a "
                                       "value for it yet" % name)
            old = getattr(o, name)
            hit = g.compare("eq", k, g.const(float(i)))
            new_a, old_a = np.asarray(value, dtype=object), np.asarray(old, dtype=object)
            if new_a.shape != old_a.shape:
                raise TraceUnsupported("assignment to attribute %r of a randomly chosen object changes its shape" % name)
            flat = [Sym(g.ite(hit, _lift(a), _lift(b))) for a, b in zip(new_a.reshape(-1), old_a.reshape(-1))]
            merged = np.empty(len(flat), dtype=object)
            merged[:] = flat
            setattr(o, name, merged.reshape(old_a.shape) if old_a.ndim else flat[0])


def _select_values(k, vals, name):
    g = _Ctx.graph
    if all(v is vals[0] for v in vals):
        return vals[0]
    if all(isinstance(v, (int, bool, np.integer, np.bool_)) for v in vals):
        return SymPickInt(g.select(k, [_lift(v) for v in vals]), k.value[0])
    if all(isinstance(v, (int, float, bool, np.integer, np.floating, np.bool_, Sym)) for v in vals):
        return Sym(g.select(k, [_lift(v) for v in vals]))
    if all(isinstance(v, np.ndarray) for v in vals):
        if len(set(v.shape for v in vals)) != 1:
            raise TraceUnsupported("attribute %r of a randomly chosen object has different shapes" % name)
        out = np.empty(vals[0].size, dtype=object)
        out[:] = [Sym(g.select(k, [_lift(v.reshape(-1)[j]) for v in vals])) for j in range(vals[0].size)]
        return out.reshape(vals[0].shape)
    if all(v is None for v in vals):
        return None
    if any(isinstance(v, (np.ndarray, int, float, Sym, str)) or v is None for v in vals):
        raise TraceUnsupported("attribute %r of a randomly chosen object mixes kinds of values" % name)
    return PickProxy(k, vals)          # objects (entity.state ...): select further down


class _Recorder(object):
    """np.random for a traced reset_world: uniform draws become U inputs, choices become picks."""

    def __init__(self, graph, forced=None):
        self.g = graph
        self.forced = dict(forced or {})      # pick number -> the value it has in THIS trace (see NeedConcretePick)
        self.draws = []           # ("uniform", lo, hi, n) | ("choice", population size)  in call order: the file's random stream
        self.n_u = 0
        self.pops = []

    def uniform(self, low=0.0, high=1.0, size=None):
        if isinstance(low, (Sym, np.ndarray)) or isinstance(high, (Sym, np.ndarray)):
            raise TraceUnsupported("np.random.uniform with array / state-dependent bounds")
        n = 1 if size is None else int(np.prod(size))
        lo, hi = float(low), float(high)
        self.draws.append(("uniform", lo, hi, n))
        vals = []
        for _ in range(n):
            u = self.g.node("U", (), (self.n_u,))
            self.n_u += 1
            # NumPy: low + (high - low) * random_sample()
            vals.append(Sym(self.g.binary("add", self.g.const(lo), self.g.binary("mul", self.g.const(hi - lo), u))))
        if size is None:
            return vals[0]
        out = np.empty(n, dtype=object)
        out[:] = vals
        return out.reshape(size)

    # rand / random / random_sample: the [0, 1) draws uniform() itself scales -- the same stream, value for value
    def random_sample(self, size=None):
        return self.uniform(0.0, 1.0, size)

    def rand(self, *dims):
        return self.uniform(0.0, 1.0, dims if dims else None)

    random = ranf = sample = random_sample

    def choice(self, a, size=None, replace=True, p=None):
        if size is not None or p is not None:
            raise TraceUnsupported("np.random.choice with size / p")
        pop = list(range(a)) if isinstance(a, (int, np.integer)) else list(a)
        if len(pop) == 0:
            raise TraceUnsupported("np.random.choice of an empty population")
        idx = len(self.pops)
        k = self.g.node("K", (), (idx,))
        self.pops.append(len(pop))
        self.draws.append(("choice", len(pop)))
        if idx in self.forced:
            return pop[self.forced[idx]]
        if all(isinstance(v, (int, np.integer)) for v in pop):
            return SymPickInt(self.g.select(k, [self.g.const(float(v)) for v in pop]), idx)
        if all(isinstance(v, (int, float, np.integer, np.floating)) for v in pop):
            return Sym(self.g.select(k, [self.g.const(float(v)) for v in pop]))
        return PickProxy(k, pop)

    def randint(self, low, high=None, size=None, dtype=int):
        if size is not None:
            raise TraceUnsupported("np.random.randint with size")
        lo, hi = (0, low) if high is None else (low, high)
        return self.choice(list(range(int(lo), int(hi))))



class _NoDraws(_Recorder):
    """np.random while observation / reward / done / benchmark_data are traced: those callbacks must not draw."""

    def _refuse(self, *a, **k):
        raise TraceUnsupported("a callback other than reset_world draws random numbers")

    uniform = choice = randint = random_sample = rand = random = ranf = sample = _refuse


class _HybridRecorder(_Recorder):
    """np.random for a reset_world that is not a formula of its draws (host_reset): its NUMBERS are drawn for real -- the
    rejection loop ends, randn / shuffle are what they are -- while its PICKS (np.random.choice / randint) stay symbolic, so that
    what the callbacks read through a picked object (`world.goal = np.random.choice(world.landmarks)`) is traced as a selection
    by a per-world pick as for any other file.  Built OUTSIDE patched_random: it keeps the real functions."""
    all_draws = True

    def __init__(self, graph, forced=None):
        _Recorder.__init__(self, graph, forced)
        self.real = {name: getattr(np.random, name) for name in ("uniform", "rand", "random", "random_sample", "ranf", "sample")}

    def uniform(self, *a, **k): return self.real["uniform"](*a, **k)
    def rand(self, *a, **k): return self.real["rand"](*a, **k)
    def random(self, *a, **k): return self.real["random"](*a, **k)
    def random_sample(self, *a, **k): return self.real["random_sample"](*a, **k)
    def ranf(self, *a, **k): return self.real["ranf"](*a, **k)
    def sample(self, *a, **k): return self.real["sample"](*a, **k)


class PickLogger(object):
    """np.random.choice / randint of a reset_world run CONCRETELY (a host reset): answered from the real stream -- choice(seq) is
    seq[randint(0, len(seq))], as NumPy's own -- or from `forced`, and logged: the picks of that world, in call order."""

    def __init__(self, forced=None):
        self.real_randint, self.log, self.forced = np.random.randint, [], None if forced is None else [int(x) for x in forced]

    def choice(self, a, size=None, replace=True, p=None):
        if size is not None or p is not None:
            raise TraceUnsupported("np.random.choice with size / p")
        pop = list(range(a)) if isinstance(a, (int, np.integer)) else list(a)
        if self.forced is not None:
            if len(self.log) >= len(self.forced):
                raise TraceUnsupported("reset_world makes more picks than were traced")
            i = self.forced[len(self.log)]
        else:
            i = int(self.real_randint(0, len(pop)))
        self.log.append(i)
        return pop[i]

    def randint(self, low, high=None, size=None, dtype=int):
        if size is not None:
            raise TraceUnsupported("np.random.randint with size")
        lo, hi = (0, low) if high is None else (low, high)
        return self.choice(list(range(int(lo), int(hi))))


@contextlib.contextmanager
def logged_picks(logger):
    saved = (np.random.choice, np.random.randint)
    np.random.choice, np.random.randint = logger.choice, logger.randint
    try:
        yield logger
    finally:
        np.random.choice, np.random.randint = saved


class _Replayer(object):
    """np.random for a CONCRETE run of reset_world with prescribed outcomes (the verification of a trace, and seeded resets)."""

    def __init__(self, uniforms, picks):
        self.u, self.k = list(uniforms), list(picks)
        self.iu = self.ik = 0

    def uniform(self, low=0.0, high=1.0, size=None):
        n = 1 if size is None else int(np.prod(size))
        r = np.array(self.u[self.iu:self.iu + n], np.float64)
        self.iu += n
        v = low + (high - low) * r
        return float(v[0]) if size is None else v.reshape(size)

    def random_sample(self, size=None):
        return self.uniform(0.0, 1.0, size)

    def rand(self, *dims):
        return self.uniform(0.0, 1.0, dims if dims else None)

    random = ranf = sample = random_sample

    def choice(self, a, size=None, replace=True, p=None):
        pop = list(range(a)) if isinstance(a, (int, np.integer)) else list(a)
        i = int(self.k[self.ik])
        self.ik += 1
        return pop[i]

    def randint(self, low, high=None, size=None, dtype=int):
        lo, hi = (0, low) if high is None else (low, high)
        return self.choice(list(range(int(lo), int(hi))))


# ---- NumPy functions whose object-dtype loops would ask symbolic comparisons for their truth (one fork per element) ---------------
def _has_sym(*xs):
    for x in xs:
        if isinstance(x, (Sym, SymBool)):
            return True
        if isinstance(x, np.ndarray) and x.dtype == object and any(isinstance(v, (Sym, SymBool)) for v in x.reshape(-1)):
            return True
        if isinstance(x, (list, tuple)) and _has_sym(*x):
            return True
    return False


def _elementwise(f, *xs):
    arrs = np.broadcast_arrays(*[np.asarray(x, dtype=object) for x in xs])
    if arrs[0].ndim == 0:
        return f(*[a.item() for a in arrs])
    out = np.empty(arrs[0].size, dtype=object)
    out[:] = [f(*vals) for vals in zip(*[a.reshape(-1) for a in arrs])]
    return out.reshape(arrs[0].shape)


def _numpy_patches():
    o_max, o_min, o_clip, o_amin, o_amax, o_where = np.maximum, np.minimum, np.clip, np.amin, np.amax, np.where

    def maximum(a, b, *args, **kw):
        if args or kw or not _has_sym(a, b):
            return o_max(a, b, *args, **kw)
        return _elementwise(lambda x, y: sym_max(x, y), a, b)

    def minimum(a, b, *args, **kw):
        if args or kw or not _has_sym(a, b):
            return o_min(a, b, *args, **kw)
        return _elementwise(lambda x, y: sym_min(x, y), a, b)

    def clip(a, a_min=None, a_max=None, *args, **kw):
        if args or kw or not _has_sym(a, a_min, a_max):
            return o_clip(a, a_min, a_max, *args, **kw)
        r = a
        if a_min is not None:
            r = maximum(r, a_min)
        if a_max is not None:
            r = minimum(r, a_max)
        return r

    def reduce(a, axis, keepdims, f):
        a = np.asarray(a, dtype=object)
        if axis is None:
            r = f(list(a.reshape(-1)))
            if keepdims:
                out = np.empty((1,) * a.ndim, dtype=object)
                out.reshape(-1)[0] = r
                return out
            return r
        if not isinstance(axis, (int, np.integer)):
            raise TraceUnsupported("a reduction of symbolic values over several axes")
        ax = int(axis) % a.ndim
        moved = np.moveaxis(a, ax, -1)
        rows = moved.reshape(-1, moved.shape[-1])
        out = np.empty(rows.shape[0], dtype=object)
        for k in range(rows.shape[0]):
            out[k] = f(list(rows[k]))
        out = out.reshape(moved.shape[:-1])
        if keepdims:
            out = np.expand_dims(out, ax)
        return out if out.ndim else out.item()

    def reducer(orig, f):
        def g(a, axis=None, *args, **kw):
            keep = kw.pop("keepdims", False) if set(kw) <= {"keepdims"} else None
            if args or kw or keep is None or not _has_sym(a):
                if keep:
                    kw["keepdims"] = keep
                return orig(a, axis, *args, **kw)
            return reduce(a, axis, keep, f)
        return g
    amin, amax = reducer(o_amin, sym_min), reducer(o_amax, sym_max)
    o_any, o_all, o_count = np.any, np.all, np.count_nonzero

    o_sort, o_argsort = np.sort, np.argsort

    def sort(a, axis=-1, *args, **kw):
        if args or kw or not _has_sym(a):
            return o_sort(a, axis, *args, **kw)
        a = np.asarray(a, dtype=object)
        if axis is None:
            a, axis = a.reshape(-1), 0
        moved = np.moveaxis(a, axis, -1)
        rows = moved.reshape(-1, moved.shape[-1])
        out = np.empty(rows.shape, dtype=object)
        for k in range(rows.shape[0]):
            out[k, :] = sorted_values(list(rows[k]))
        return np.moveaxis(out.reshape(moved.shape), -1, axis)

    def argsort(a, axis=-1, *args, **kw):
        if args or kw or not _has_sym(a) or np.ndim(a) != 1:
            return o_argsort(a, axis, *args, **kw)
        xs = list(np.asarray(a, dtype=object))
        return np.array(sorted(range(len(xs)), key=lambda i: xs[i]), dtype=np.int64)          # (the ORDER is a decision: forks)

    def arg_extreme(orig, op):
        def g(a, axis=None, *args, **kw):
            if axis is not None or args or kw or not _has_sym(a):
                return orig(a, axis, *args, **kw)
            return first_extreme(np.asarray(a, dtype=object).reshape(-1), op)
        return g

    def count_nonzero(a, axis=None, **kw):
        if kw or not _has_sym(a):
            return o_count(a, axis, **kw)
        return reduce(a, axis, False, lambda xs: sum(Sym(_Ctx.graph.as_float(_truthy(x))) for x in xs))

    def where(c, *rest):
        if len(rest) != 2 or not _has_sym(c):
            return o_where(c, *rest)

        def pick(t, x, y):
            if isinstance(t, (bool, np.bool_)):
                return x if t else y
            return _select_any(t, x, y)
        return _elementwise(pick, c, rest[0], rest[1])
    maximum.reduce = lambda a, axis=0, **kw: amax(a, axis, **kw)          # (np.minimum.reduce(d): the ufunc's method)
    minimum.reduce = lambda a, axis=0, **kw: amin(a, axis, **kw)
    o_isclose, o_allclose = np.isclose, np.allclose

    def isclose(a, b, rtol=1e-05, atol=1e-08, equal_nan=False):
        if equal_nan or not _has_sym(a, b):
            return o_isclose(a, b, rtol=rtol, atol=atol, equal_nan=equal_nan)
        return _elementwise(lambda x, y: abs(Sym(_lift(x)) - Sym(_lift(y))) <= atol + rtol * abs(Sym(_lift(y))), a, b)

    def allclose(a, b, rtol=1e-05, atol=1e-08, equal_nan=False):
        if equal_nan or not _has_sym(a, b):
            return o_allclose(a, b, rtol=rtol, atol=atol, equal_nan=equal_nan)
        r = isclose(a, b, rtol, atol)
        return sym_all(list(np.asarray(r, dtype=object).reshape(-1)))
    o_array_equal = np.array_equal

    def array_equal(a, b, *args, **kw):
        if args or kw or not _has_sym(a, b):
            return o_array_equal(a, b, *args, **kw)
        aa, bb = np.asarray(a, dtype=object), np.asarray(b, dtype=object)
        if aa.shape != bb.shape:
            return False
        return sym_all([x == y for x, y in zip(aa.reshape(-1), bb.reshape(-1))])
    out = {"maximum": maximum, "minimum": minimum, "clip": clip, "amin": amin, "amax": amax, "min": amin, "max": amax, "where": where,
           "isclose": isclose, "allclose": allclose, "array_equal": array_equal,
           "any": reducer(o_any, sym_any), "all": reducer(o_all, sym_all), "count_nonzero": count_nonzero,
           "argmin": arg_extreme(np.argmin, "min"), "argmax": arg_extreme(np.argmax, "max"), "sort": sort, "argsort": argsort}

    # ---- elementwise functions: NumPy's object loops call a METHOD of each element (x.sqrt()), which the plain Python floats that
    # share an array with symbolic values do not have; np.sign orders its argument against 0 (one fork per element)
    def unary(name, method):
        orig = getattr(np, name)

        def one(v):
            if isinstance(v, SymBool):
                v = v._f()
            if isinstance(v, Sym):
                m = getattr(v, method, None)
                if m is None:
                    raise TraceUnsupported("np.%s of a state-dependent value" % name)
                return m()
            return orig(v)

        def f(x, *args, **kw):
            if args or kw or not (isinstance(x, (Sym, SymBool)) or (isinstance(x, np.ndarray) and x.dtype == object)
                                  or (isinstance(x, (list, tuple)) and _has_sym(x))):
                return orig(x, *args, **kw)
            if not _has_sym(x):                       # numbers that only happen to sit in an object array
                return orig(np.asarray(x, dtype=np.float64)).astype(object)
            return _elementwise(one, x)
        f.__name__ = name
        return f
    for name, method in [("sqrt", "sqrt"), ("exp", "exp"), ("log", "log"), ("tanh", "tanh"), ("sin", "sin"), ("cos", "cos"),
                         ("floor", "floor"), ("ceil", "ceil"), ("rint", "rint"), ("trunc", "trunc"), ("sign", "sign"),
                         ("fabs", "__abs__"), ("square", "square"), ("reciprocal", "reciprocal")] + \
                        [("isfinite", "isfinite"), ("isnan", "isnan"), ("isinf", "isinf")] + \
                        [(n, None) for n in ("arctan", "arcsin", "arccos", "sinh", "cosh", "log1p", "expm1", "log2", "log10", "cbrt",
                                             "degrees", "radians")]:
        out[name] = unary(name, method or "_no_such_method_")

    return out


_CTOR_NAMES = (("zeros", False), ("ones", False), ("empty", False), ("full", False),
               ("zeros_like", True), ("ones_like", True), ("empty_like", True), ("full_like", True))


def _ctor_patches():
    """`o = np.zeros(4); o[:2] = agent.state.p_pos` needs an array that can hold symbolic values: np.zeros / ones / empty / full
    (_like) giving object arrays where the file did not ask for another dtype.  NOT patched into the numpy module: compiled
    library code (numpy.random's Cython, scipy) fills what np.empty(n) returns through a C pointer and would write doubles over
    object pointers.  Only the FILE sees them -- injected_builtins binds its `np` to a proxy (and its `from numpy import zeros`
    names to these functions) while it is traced."""
    def ctor(name, like):
        orig = getattr(np, name)

        def f(*args, **kw):
            dt = kw.get("dtype", None)
            if dt is sym_int:
                kw["dtype"] = dt = int
            if dt is None and like and args and isinstance(args[0], np.ndarray) and args[0].dtype.kind != "f":
                return orig(*args, **kw)              # (zeros_like an integer / object array keeps its kind)
            positional_dtype = len(args) > (2 if name in ("full", "full_like") else 1) and not like
            if positional_dtype or not (dt is None or dt is sym_float or dt is object):
                return orig(*args, **kw)
            kw.pop("dtype", None)
            r = orig(*args, **kw)
            return r.astype(object) if r.dtype.kind == "f" else r
        f.__name__ = name
        return f
    out = {name: ctor(name, like) for name, like in _CTOR_NAMES}

    def with_dtype(name):          # np.array(parts, dtype=np.float32), np.asarray(row, dtype=np.float32)
        orig = getattr(np, name)

        def f(a, dtype=None, *args, **kw):
            if dtype is None or args or not _has_sym(a):
                return orig(a, dtype, *args, **kw) if dtype is not None else orig(a, *args, **kw)
            return _as_dtype(orig(a, dtype=object, **kw), dtype)
        f.__name__ = name
        return f
    out["array"], out["asarray"] = with_dtype("array"), with_dtype("asarray")

    def scalar(orig, conv):
        class _Scalar(orig):          # np.float32(x) of a symbolic x; isinstance(v, np.float32) keeps working
            def __new__(cls, x=0, *args):
                if isinstance(x, (Sym, SymBool)) and not args:
                    return conv(Sym(_lift(x)))
                if isinstance(x, np.ndarray) and x.dtype == object and _has_sym(x) and not args:
                    return _as_dtype(x, orig)
                return orig(x, *args)
        _Scalar.__name__ = orig.__name__
        return _Scalar
    out["float32"], out["float64"] = scalar(np.float32, lambda x: x.f32()), scalar(np.float64, lambda x: x)
    return out


class _NumpyProxy(object):
    """What the traced file's `np` is bound to: numpy, with the constructors of _ctor_patches."""

    def __init__(self, real, over):
        object.__setattr__(self, "_real", real)
        object.__setattr__(self, "_over", over)

    def __getattr__(self, name):
        over = object.__getattribute__(self, "_over")
        if name in over:
            return over[name]
        return getattr(object.__getattribute__(self, "_real"), name)

    def __setattr__(self, name, value):
        setattr(object.__getattribute__(self, "_real"), name, value)

    def __dir__(self):
        return dir(object.__getattribute__(self, "_real"))


# ---- the math module: its functions take floats (a symbolic value would be asked for its __float__) -------------------------------
def _math_wrappers():
    def unary(name, sym):
        orig = getattr(math, name)

        def f(x):
            return sym(x if isinstance(x, Sym) else Sym(_Ctx.graph.as_float(x.n))) if isinstance(x, (Sym, SymBool)) else orig(x)
        f.__name__ = name
        return f

    def binary(name, sym):
        orig = getattr(math, name)

        def f(x, y):
            if isinstance(x, (Sym, SymBool)) or isinstance(y, (Sym, SymBool)):
                return sym(Sym(_lift(x)), Sym(_lift(y)))
            return orig(x, y)
        f.__name__ = name
        return f
    return {"sqrt": unary("sqrt", lambda x: x.sqrt()), "exp": unary("exp", lambda x: x.exp()), "log": unary("log", lambda x: x.log()),
            "tanh": unary("tanh", lambda x: x.tanh()), "sin": unary("sin", lambda x: x.sin()), "cos": unary("cos", lambda x: x.cos()),
            "fabs": unary("fabs", lambda x: abs(x)), "atan2": binary("atan2", lambda x, y: x.arctan2(y)),
            "hypot": binary("hypot", lambda x, y: (x * x + y * y).sqrt()), "pow": binary("pow", lambda x, y: x ** (y.n.value if y.n.op == "const" else y))}


_MATH_WRAPPERS = None


@contextlib.contextmanager
def patched_math():
    """math.sqrt / exp / log / tanh / sin / cos / fabs / atan2 / hypot / pow accept symbolic values while a file is traced (and
    while its source is re-executed for predication: `from math import sqrt` then binds the wrapper, which is the original
    function for ordinary numbers)."""
    global _MATH_WRAPPERS
    if _MATH_WRAPPERS is None:
        _MATH_WRAPPERS = _math_wrappers()
    saved = {name: getattr(math, name) for name in _MATH_WRAPPERS}
    if any(saved[name] is _MATH_WRAPPERS[name] for name in saved):      # (already patched: nested use)
        yield
        return
    try:
        for name, f in _MATH_WRAPPERS.items():
            setattr(math, name, f)
        yield
    finally:
        for name, f in saved.items():
            setattr(math, name, f)


_RANDOM_NAMES = ("uniform", "choice", "randint", "rand", "random", "random_sample", "ranf", "sample")
_RANDOM_REFUSED = ("randn", "normal", "shuffle", "permutation", "standard_normal",
                   "exponential", "beta", "gamma", "seed", "binomial", "poisson")


@contextlib.contextmanager
def patched_random(impl):
    """np.random.{uniform, rand, random, random_sample, choice, randint} answered by `impl` (a _Recorder / _Replayer); everything
    else that draws raises."""
    saved = {}
    for name in _RANDOM_NAMES + _RANDOM_REFUSED:
        if hasattr(np.random, name):
            saved[name] = getattr(np.random, name)
    patches = _numpy_patches() if isinstance(impl, _Recorder) else {}      # (a symbolic run: max / min / clip / where without forks)
    saved_np = {name: getattr(np, name) for name in patches}
    mathctx = patched_math() if isinstance(impl, _Recorder) else contextlib.nullcontext()
    mathctx.__enter__()
    try:
        for name, f in patches.items():
            setattr(np, name, f)
        for name in _RANDOM_NAMES:
            setattr(np.random, name, getattr(impl, name))
        for name in (() if getattr(impl, "all_draws", False) else _RANDOM_REFUSED):
            if name in saved:
                def refuse(*a, _n=name, **k):
                    raise TraceUnsupported("np.random.%s in a traced callback" % _n)
                setattr(np.random, name, refuse)
        yield
    finally:
        for name, f in saved.items():
            setattr(np.random, name, f)
        for name, f in saved_np.items():
            setattr(np, name, f)
        mathctx.__exit__(None, None, None)


@contextlib.contextmanager
def injected_builtins(scenario):
    """min / max / any / all / float / int / round / sorted -- and numpy's array constructors (_ctor_patches) -- as the file's module
    sees them while it is traced."""
    import types
    pkg = __name__.rsplit(".", 1)[0]
    spaces = []          # the global namespaces the scenario's methods (and its base classes') resolve names in
    for klass in type(scenario).__mro__:
        if klass.__module__ == "builtins" or klass.__module__.startswith(pkg + "."):
            continue
        for f in klass.__dict__.values():
            f = getattr(f, "__func__", f)
            if isinstance(f, types.FunctionType) and not any(f.__globals__ is d for d in spaces):
                spaces.append(f.__globals__)
    # ... and the user's own helper modules next to the file (`import helpers`, `from helpers import dist`): the same names there
    try:
        import os
        home = os.path.dirname(os.path.abspath(source_file(type(scenario))))
    except Exception:
        home = None
    if home:
        def neighbour(d):
            f = d.get("__file__")
            return isinstance(f, str) and os.path.abspath(f).startswith(home + os.sep) and not d.get("__name__", "").startswith(pkg + ".")
        for d in list(spaces):
            for v in list(d.values()):
                g = v.__dict__ if isinstance(v, types.ModuleType) else getattr(getattr(v, "__func__", v), "__globals__", None)
                if isinstance(g, dict) and neighbour(g) and not any(g is x for x in spaces):
                    spaces.append(g)
    saved = [(d, name, d.get(name, _MISSING)) for d in spaces for name in _INJECTED]
    ctors = _ctor_patches()
    by_original = {getattr(np, name): f for name, f in ctors.items()}
    proxy = _NumpyProxy(np, ctors)
    for d in spaces:          # the file's `np` (however it is spelt) and its `from numpy import zeros` names
        for name, v in list(d.items()):
            if name in _INJECTED:
                continue
            if v is np:
                saved.append((d, name, v))
            elif callable(v) and not isinstance(v, type):
                try:
                    hit = v in by_original
                except TypeError:
                    hit = False
                if hit:
                    saved.append((d, name, v))
    try:
        for d in spaces:
            d.update(_INJECTED)
        for d, name, old in saved[len(spaces) * len(_INJECTED):]:
            d[name] = proxy if old is np else by_original[old]
        yield
    finally:
        for d, name, old in saved:
            if old is _MISSING:
                d.pop(name, None)
            else:
                d[name] = old


_MISSING = object()


# ---- the trace of one scenario ----------------------------------------------------------------------------------------------
class _NotTraced(object):
    """What stands where a callback must not look while it is traced: any use says what was read."""

    def __init__(self, what):
        object.__setattr__(self, "_what", what)

    def _refuse(self, *a, **k):
        raise TraceUnsupported("a callback reads %s: only the world's state is traced (docs/TRACER.md)" % object.__getattribute__(self, "_what"))

    __getattr__ = __getitem__ = __iter__ = __len__ = __array__ = __float__ = __bool__ = __neg__ = __abs__ = _refuse
    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __truediv__ = __rtruediv__ = __pow__ = __rpow__ = _refuse
    __lt__ = __le__ = __gt__ = __ge__ = __eq__ = __ne__ = _refuse
    __hash__ = None


def _obj_vec(nodes):
    out = np.empty(len(nodes), dtype=object)
    out[:] = [Sym(n) for n in nodes]
    return out


def _flatten(raw):
    """One path's return value of `observation` -> list of Nodes."""
    if raw is None:
        raise TraceUnsupported("observation returned None")
    arr = np.asarray(raw, dtype=object).reshape(-1) if not isinstance(raw, (Sym, SymBool)) else [raw]
    return [_lift(x) for x in arr]


def _scalar(raw):
    if isinstance(raw, np.ndarray):
        if raw.size != 1:
            raise TraceUnsupported("reward returned an array of %d values" % raw.size)
        raw = raw.reshape(-1)[0]
    if raw is None:
        raise TraceUnsupported("reward returned None")
    return [_lift(raw)]


def _truth(raw):
    if isinstance(raw, SymBool):
        return [raw.n]
    if isinstance(raw, (bool, np.bool_)):
        return [_Ctx.graph.const(bool(raw))]
    if isinstance(raw, Sym):
        return [_Ctx.graph.compare("ne", raw.n, _Ctx.graph.const(0.0))]
    if isinstance(raw, (int, float, np.integer, np.floating)):
        return [_Ctx.graph.const(bool(raw))]
    raise TraceUnsupported("done returned %r" % type(raw).__name__)


def _describe_info(raw):
    """The shape of what benchmark_data returns, from a CONCRETE call: ("s", kind) a number, ("a", n, kind) an array of n numbers,
    ("t", [..]) a tuple, ("none",) an empty dict; kind "i" (ints / bools: delivered as int32) or "f"."""
    if isinstance(raw, dict) and not raw:
        return ("none",)
    if isinstance(raw, (bool, np.bool_, int, np.integer)):
        return ("s", "i")
    if isinstance(raw, (float, np.floating)):
        return ("s", "f")
    if isinstance(raw, np.ndarray):
        if raw.ndim == 0:
            return ("s", "i" if raw.dtype.kind in "iub" else "f")
        return ("a", int(raw.size), "i" if raw.dtype.kind in "iub" else "f")
    if isinstance(raw, (tuple, list)):
        return ("t", [_describe_info(x) for x in raw])
    raise TraceUnsupported("benchmark_data returns %r" % type(raw).__name__)


def _info_width(desc):
    return {"none": 0, "s": 1}.get(desc[0], None) if desc[0] in ("none", "s") else (desc[1] if desc[0] == "a" else sum(_info_width(d) for d in desc[1]))


def _flatten_info(raw, desc):
    """One path's return value of benchmark_data -> list of Nodes, in the order of `desc`."""
    if desc[0] == "none":
        if not (isinstance(raw, dict) and not raw):
            raise TraceUnsupported("benchmark_data returns different structures on different paths")
        return []
    if desc[0] == "s":
        if isinstance(raw, np.ndarray):
            raw = raw.reshape(-1)[0]
        return [_lift(raw)]
    if desc[0] == "a":
        a = np.asarray(raw, dtype=object).reshape(-1)
        if a.size != desc[1]:
            raise TraceUnsupported("benchmark_data returns arrays of different sizes on different paths")
        return [_lift(x) for x in a]
    if not isinstance(raw, (tuple, list)) or len(raw) != len(desc[1]):
        raise TraceUnsupported("benchmark_data returns different structures on different paths")
    return [n for x, d in zip(raw, desc[1]) for n in _flatten_info(x, d)]


class Traced(object):
    host_reset = None          # why reset_world is not a traced program but the file's own Python, run per world (None: it is traced)
    params = ()                # uniform draws of reset_world that the CALLBACKS read (hidden per-world numbers): their U indices; they
    #                            occupy the pick slots after the real picks -- pops[n_picks + j] == PARAM_POP, value k = floor(u * 2^24)
    n_picks = None             # how many of `pops` are np.random.choice picks (None: all of them)

    def real_picks(self):
        return len(self.pops) if self.n_picks is None else self.n_picks

    """What tracing a reference-style scenario yields: per-agent graphs over the state + the reset program.

      obs[i]        list of Nodes: agent i's observation row, column by column
      rew[i]        Node: agent i's reward
      done[i]       Node (bool) or None
      reset_pos     [E][2] Nodes over U (uniform draws in [0,1)) and K (picks): reset_world's positions
      reset_vel     [E][2] Nodes;  reset_c: [A][dim_c] Nodes
      draws         reset_world's random stream, in call order: ("uniform", lo, hi, n) | ("choice", n)
      pops          population size of every pick
    """

    def __init__(self):
        self.graph = Graph()


# ---- a trace as plain data (JSON): cached beside the compiled images; test fixtures of the reference's own nine files ----------
_ENT_KEYS = ("name", "size", "movable", "collide", "max_speed", "accel", "initial_mass", "density")
_AGT_KEYS = ("silent", "blind", "u_noise", "c_noise", "u_range")
_WLD_KEYS = ("dim_c", "dim_p", "dim_color", "dt", "damping", "contact_force", "contact_margin")


def to_dict(t):
    g = t.graph
    nodes = sorted(g.nodes.values(), key=lambda n: n.uid)
    assert [n.uid for n in nodes] == list(range(len(nodes)))
    w = t.world

    def ent(e, agent):
        d = {k: getattr(e, k, None) for k in _ENT_KEYS}
        if agent:
            d.update({k: getattr(e, k, None) for k in _AGT_KEYS})
        return d
    return {"format": 1,
            "nodes": [[n.op, [a.uid for a in n.args], list(n.value) if isinstance(n.value, tuple) else n.value] for n in nodes],
            "obs": [[n.uid for n in row] for row in t.obs], "rew": [n.uid for n in t.rew],
            "done": [None if d is None else d.uid for d in t.done],
            "reset_pos": [[n.uid for n in e] for e in t.reset_pos], "reset_vel": [[n.uid for n in e] for e in t.reset_vel],
            "reset_c": [[n.uid for n in a] for a in t.reset_c],
            "draws": [list(d) for d in t.draws], "pops": list(t.pops), "n_u": t.n_u, "A": t.A, "E": t.E, "dim_c": t.dim_c,
            "info": None if getattr(t, "info", None) is None else [[n.uid for n in row] for row in t.info],
            "info_desc": getattr(t, "info_desc", None),
            "predicated": bool(getattr(t, "predicated", False)), "host_reset": getattr(t, "host_reset", None),
            "params": list(getattr(t, "params", ())), "n_picks": getattr(t, "n_picks", None),
            "collaborative": bool(t.collaborative), "paths": t.paths, "enumerated": list(getattr(t, "enumerated", [])),
            "world": {k: getattr(w, k) for k in _WLD_KEYS},
            "discrete_action": getattr(w, "discrete_action", None),
            "agents": [ent(a, True) for a in w.agents], "landmarks": [ent(l, False) for l in w.landmarks]}


def from_dict(d):
    """The inverse of to_dict: a Traced whose `world` is a world of compat.core objects carrying the recorded constants."""
    from .compat import core as ccore
    if d.get("format") != 1:
        raise TraceUnsupported("trace data of another format")
    t = Traced()
    g = t.graph
    nodes = []
    for op, args, value in d["nodes"]:
        n = g.node(op, tuple(nodes[a] for a in args), tuple(value) if isinstance(value, list) else value)
        if n.uid != len(nodes):
            raise TraceUnsupported("trace data is not in creation order")
        nodes.append(n)
    t.obs = [[nodes[u] for u in row] for row in d["obs"]]
    t.rew = [nodes[u] for u in d["rew"]]
    t.done = [None if u is None else nodes[u] for u in d["done"]]
    t.reset_pos = [[nodes[u] for u in e] for e in d["reset_pos"]]
    t.reset_vel = [[nodes[u] for u in e] for e in d["reset_vel"]]
    t.reset_c = [[nodes[u] for u in a] for a in d["reset_c"]]
    def desc(x):          # JSON lists back into the descriptors of _describe_info: ("t", [descriptors]) | ("s", kind) | ("a", n, kind) | ("none",)
        return ("t", [desc(y) for y in x[1]]) if x[0] == "t" else tuple(x)
    t.info = None if d.get("info") is None else [[nodes[u] for u in row] for row in d["info"]]
    t.info_desc = None if d.get("info_desc") is None else [desc(x) for x in d["info_desc"]]
    t.draws = [tuple(x) for x in d["draws"]]
    t.pops, t.n_u, t.A, t.E, t.dim_c = list(d["pops"]), d["n_u"], d["A"], d["E"], d["dim_c"]
    t.collaborative, t.paths, t.enumerated = d["collaborative"], d["paths"], d["enumerated"]
    t.predicated = bool(d.get("predicated", False))
    t.host_reset = d.get("host_reset")
    t.params, t.n_picks = list(d.get("params") or ()), d.get("n_picks")
    w = ccore.World()
    for k, v in d["world"].items():
        setattr(w, k, v)
    if d.get("discrete_action") is not None:
        w.discrete_action = d["discrete_action"]
    w.collaborative = t.collaborative
    for spec, klass, dst in [(x, ccore.Agent, w.agents) for x in d["agents"]] + [(x, ccore.Landmark, w.landmarks) for x in d["landmarks"]]:
        e = klass()
        for k, v in spec.items():
            setattr(e, k, v)
        dst.append(e)
    t.world = w
    return t


def _entity_lists(world):
    return list(world.agents), list(world.agents) + list(world.landmarks)


def _trace_once(scenario, t, forced, want_done, max_paths, want_info=False):
    """One pass: make_world, reset_world with np.random recorded (picks in `forced` take their given value), the callbacks over
    the symbolic state.  -> dict of node lists."""
    g = t.graph
    world = scenario.make_world()            # (make_world ends with a concrete reset_world: the reference's own contract)
    agents, ents = _entity_lists(world)
    A, E, dp, dc = len(agents), len(ents), int(world.dim_p), int(world.dim_c)
    if dp != 2:
        raise TraceUnsupported("dim_p = %d" % dp)
    if any(a.action_callback is not None for a in agents):
        raise TraceUnsupported("scripted agents (action_callback)")
    if any(l.movable for l in world.landmarks):
        raise TraceUnsupported("movable landmarks")
    if any(a.u_noise or (a.c_noise and not a.silent) for a in agents):
        raise TraceUnsupported("action / communication noise")
    t.world, t.A, t.E, t.dim_c = world, A, E, dc
    info_desc = None
    if want_info:      # what benchmark_data returns (numbers? tuples? ints?), from a concrete call on the world make_world left
        for a in agents:
            a.action.c = np.zeros(dc)
            a.action.u = np.zeros(dp)
        info_desc = [_describe_info(scenario.benchmark_data(a, world)) for a in agents]
    # ---- reset_world, symbolically ------------------------------------------------------------------------------------------
    def attempt(rec):
        with patched_random(rec), injected_builtins(scenario):
            _, conds, _ = Tracer(g, 1, max_decisions=64)._run(lambda: scenario.reset_world(world), [])
        if conds:      # reset_world branches on what it drew: on a pick -> one trace per value of that pick; on anything else: not modelled
            ks = sorted(n.value[0] for n in topo(conds[:1]) if n.op == "K")
            if ks and "U" not in inputs_of(conds[:1]):
                raise NeedConcretePick(ks[0])
            raise TraceUnsupported("reset_world branches on a random number it drew")
    rec = _Recorder(g, forced)
    host_reset = None
    try:
        attempt(rec)
    except NeedConcretePick:
        raise
    except Exception as e:
        # A reset_world that is not a FORMULA of its draws -- rejection sampling (`while too_close: draw again`), shuffles, normal
        # draws -- is no reason to give up the callbacks: it stays what it is, Python run per world at reset time (host_reset), and
        # only observation / reward / done are traced.  Its picks are still followed (_HybridRecorder); whatever else it leaves
        # outside the state vectors that the callbacks then read makes the verification fail, and the file falls back as before.
        if _Ctx.need_pick is not None:
            raise
        host_reset = "%s: %s" % (type(e).__name__, e) if not isinstance(e, TraceUnsupported) else str(e)
        rec = _HybridRecorder(g, forced)
        try:
            attempt(rec)
        except NeedConcretePick:
            raise
        except Exception:
            if _Ctx.need_pick is not None:
                raise
            rec = _Recorder(g, forced)           # (not even with its numbers drawn for real: no picks, everything concrete)
            scenario.reset_world(world)          # (the world is whole again; _trace restores the caller's np.random stream)
    out = {"draws": rec.draws, "pops": rec.pops, "n_u": rec.n_u, "host_reset": host_reset}

    def vec(v, n, what):
        if v is None:
            raise TraceUnsupported("%s is None after reset_world" % what)
        a = np.asarray(v, dtype=object).reshape(-1)
        if a.size != n:
            raise TraceUnsupported("%s has %d components, expected %d" % (what, a.size, n))
        return [_lift(x) for x in a]
    out["reset_pos"] = [x for k, e in enumerate(ents) for x in vec(e.state.p_pos, 2, "p_pos of %s" % (e.name or "entity %d" % k))]
    out["reset_vel"] = [x for e in ents for x in (vec(e.state.p_vel, 2, "p_vel") if e.state.p_vel is not None else [g.const(0.0)] * 2)]
    out["reset_c"] = [x for a in agents for x in (vec(a.state.c, dc, "state.c") if (dc and a.state.c is not None) else [g.const(0.0)] * dc)]
    # (host_reset: the constants of ONE concrete reset -- placeholders, never evaluated)
    # ---- the state as inputs -------------------------------------------------------------------------------------------------
    for k, e in enumerate(ents):
        e.state.p_pos = _obj_vec([g.node("P", (), (k, 0)), g.node("P", (), (k, 1))])
        if k < A and e.movable:
            e.state.p_vel = _obj_vec([g.node("V", (), (k, 0)), g.node("V", (), (k, 1))])
        else:
            e.state.p_vel = np.zeros(2)          # (never integrated: core.py:160)
    for i, a in enumerate(agents):
        # core.py:171-177: a silent agent's utterance is zeros; a speaking one's is its last communication action
        a.state.c = np.zeros(dc) if (a.silent or dc == 0) else _obj_vec([g.node("C", (), (i, c)) for c in range(dc)])
        a.action.u = _NotTraced("agent.action.u (the action of the current step)")
        a.action.c = _NotTraced("agent.action.c (the action of the current step)")
    # ---- the callbacks, every path -------------------------------------------------------------------------------------------
    out["obs"], out["rew"], out["done"] = [], [], []
    paths = {"obs": [], "rew": [], "done": []}
    if want_info:
        out["info"] = []
    with patched_random(_NoDraws(g)), injected_builtins(scenario):
        for i, a in enumerate(agents):
            tr = Tracer(g, max_paths)
            out["obs"].append(tr.explore(lambda: scenario.observation(a, world), _flatten))
            paths["obs"].append(tr.paths)
            tr = Tracer(g, max_paths)
            out["rew"].append(tr.explore(lambda: scenario.reward(a, world), _scalar))
            paths["rew"].append(tr.paths)
            if want_done and hasattr(scenario, "done"):
                tr = Tracer(g, max_paths)
                out["done"].append(tr.explore(lambda: scenario.done(a, world), _truth))
                paths["done"].append(tr.paths)
            else:
                out["done"].append([])
            if want_info:
                tr = Tracer(g, max_paths)
                out["info"].append(tr.explore(lambda: scenario.benchmark_data(a, world), lambda raw: _flatten_info(raw, info_desc[i])))
    out["info_desc"] = info_desc
    out["paths"] = paths
    out["collaborative"] = bool(getattr(world, "collaborative", False))
    return out


_MAX_ENUMERATED = 64      # traces per scenario when picks have to be enumerated (product of their population sizes)


def trace(scenario, want_done=False, max_paths=None, predicate=True, want_info=False):
    """Trace `scenario` (a reference-style Scenario object: make_world(self), reset_world(self, world), NumPy callbacks).
    Raises TraceUnsupported when the file is outside what the tracer models.  predicate: trace a twin of the scenario whose
    value-only control flow does not fork (predicated_twin); where that twin cannot be built or traced, the scenario itself."""
    if predicate:
        try:
            t = _trace(predicated_twin(scenario), want_done, max_paths, want_info)
            t.predicated = True
            return t
        except TraceUnsupported as e:
            first = e
        try:
            t = _trace(scenario, want_done, max_paths, want_info)
        except TraceUnsupported as e:
            raise TraceUnsupported(str(e) if str(e) == str(first) else "%s (with predicated control flow: %s)" % (e, first))
        t.predicated = False
        return t
    t = _trace(scenario, want_done, max_paths, want_info)
    t.predicated = False
    return t


def _trace(scenario, want_done=False, max_paths=None, want_info=False):
    import itertools
    t = Traced()
    g = t.graph
    keep_g, _Ctx.graph = _Ctx.graph, g
    rng_state = np.random.get_state()
    try:
        enumerated, pops = [], None
        while True:
            _Ctx.need_pick = None
            try:
                combos = list(itertools.product(*[range(pops[k]) for k in enumerated])) if enumerated else [()]
                runs = {c: _trace_once(scenario, t, dict(zip(enumerated, c)), want_done, max_paths, want_info) for c in combos}
                break
            except Exception as e:
                if not isinstance(e, NeedConcretePick) and _Ctx.need_pick is None:
                    raise
                need = e.k if isinstance(e, NeedConcretePick) else _Ctx.need_pick
                _Ctx.need_pick = None
                e = NeedConcretePick(need)
                if e.k in enumerated:
                    raise TraceUnsupported("pick %d is needed concretely although it is fixed" % e.k)
                if pops is None:
                    pops = _pick_populations(scenario, g)
                enumerated.append(e.k)
                n = 1
                for k in enumerated:
                    n *= pops[k]
                if n > _MAX_ENUMERATED:
                    raise TraceUnsupported("picks %s are used as concrete values: %d traces" % (enumerated, n))
        first = runs[combos[0]]
        for r in runs.values():
            if r["draws"] != first["draws"] or [len(o) for o in r["obs"]] != [len(o) for o in first["obs"]]:
                raise TraceUnsupported("reset_world's random stream (or a row width) depends on a pick")

        def merged(key, sub=None):
            """select over the enumerated picks, innermost = last enumerated"""
            def rec(prefix, depth):
                if depth == len(enumerated):
                    r = runs[tuple(prefix)][key]
                    return r if sub is None else r[sub]
                k = enumerated[depth]
                branches = [rec(prefix + [v], depth + 1) for v in range(pops[k])]
                kn = g.node("K", (), (k,))
                return [g.select(kn, [b[j] for b in branches]) for j in range(len(branches[0]))]
            return rec([], 0)
        t.draws, t.pops, t.n_u = first["draws"], first["pops"], first["n_u"]
        t.host_reset = first.get("host_reset")
        A, E, dc = t.A, t.E, t.dim_c
        flat = merged("reset_pos")
        t.reset_pos = [flat[2 * e:2 * e + 2] for e in range(E)]
        flat = merged("reset_vel")
        t.reset_vel = [flat[2 * e:2 * e + 2] for e in range(E)]
        flat = merged("reset_c")
        t.reset_c = [flat[dc * i:dc * i + dc] for i in range(A)]
        t.obs = [merged("obs", i) for i in range(A)]
        t.rew = [merged("rew", i)[0] for i in range(A)]
        t.done = [(merged("done", i)[0] if first["done"][i] else None) for i in range(A)]
        t.info, t.info_desc = None, None
        if want_info:
            if any(r["info_desc"] != first["info_desc"] for r in runs.values()):
                raise TraceUnsupported("the structure of benchmark_data depends on a pick")
            t.info = [merged("info", i) for i in range(A)]
            t.info_desc = first["info_desc"]
        t.paths = first["paths"]
        # B worlds share ONE set of physics constants (they step in one launch): a make_world that randomises them per call cannot
        # be batched -- per-world randomness belongs in reset_world (refstyle.RefScenarioAdapter refuses the same)
        np.random.seed(12345)
        other = scenario.make_world()

        def consts(w):
            return ([tuple(getattr(e, k, None) for k in _ENT_KEYS[1:]) for e in list(w.agents) + list(w.landmarks)],
                    [tuple(getattr(a, k, None) for k in _AGT_KEYS) for a in w.agents], tuple(getattr(w, k) for k in _WLD_KEYS))
        if consts(other) != consts(t.world):
            raise TraceUnsupported("make_world() builds worlds with different entity counts or physics constants from call to call")
        t.enumerated = list(enumerated)
        t.collaborative = first["collaborative"]
        # A random NUMBER reset_world drew and kept outside the state vectors (`world.goal_pos = np.random.uniform(-1, 1, 2)`: a goal that
        # is coordinates, not an entity) which the callbacks read: a per-world parameter.  It travels in a pick slot -- a "pick" among
        # 2^24 values, k = floor(u * 2^24), read back as k * 2^-24 (what a float32 uniform draw is) -- so that seeded resets, device
        # restarts and the kernel's K accessor serve it like any pick.
        groups = [[n for row in t.obs for n in row], list(t.rew), [d for d in t.done if d is not None], [n for row in (t.info or []) for n in row]]
        used = sorted({n.value[0] for roots in groups for n in topo(roots) if n.op == "U"})
        if used:
            if len(t.pops) + len(used) > MAX_SLOTS:
                raise TraceUnsupported("the callbacks read %d random numbers reset_world drew and kept outside the state, next to %d picks "
                                       "(at most %d per-world slots)" % (len(used), len(t.pops), MAX_SLOTS))
            t.n_picks, t.params = len(t.pops), list(used)
            slot = {}
            for j, u in enumerate(used):
                k = g.node("K", (), (t.n_picks + j,))
                slot[g.node("U", (), (u,)).uid] = g.binary("mul", k, g.const(1.0 / PARAM_POP))
            t.pops = list(t.pops) + [PARAM_POP] * len(used)
            memo = dict(slot)
            for n in topo([n for roots in groups for n in roots]):          # (arguments come before their users)
                if n.uid in memo:
                    continue
                args = tuple(memo[a.uid] for a in n.args)
                memo[n.uid] = n if all(a is b for a, b in zip(args, n.args)) else g.node(n.op, args, n.value)
            t.obs = [[memo[n.uid] for n in row] for row in t.obs]
            t.rew = [memo[n.uid] for n in t.rew]
            t.done = [None if d is None else memo[d.uid] for d in t.done]
            if t.info is not None:
                t.info = [[memo[n.uid] for n in row] for row in t.info]
        return t
    except TraceUnsupported:
        raise
    except Exception as e:       # the file did something a symbolic value cannot do: that is a reason to fall back, not a crash
        raise TraceUnsupported("%s: %s" % (type(e).__name__, e))
    finally:
        _Ctx.graph = keep_g
        np.random.set_state(rng_state)


def _pick_populations(scenario, g):
    """Population size of every np.random.choice of reset_world, found by a pass in which every pick is answered concretely."""
    def sizes(base):
        class _Sizes(base):
            def choice(self, a, size=None, replace=True, p=None):
                pop = list(range(a)) if isinstance(a, (int, np.integer)) else list(a)
                self.pops.append(len(pop))
                return pop[0]
        rec = _Sizes(g)
        with patched_random(rec):
            world = scenario.make_world()
            rec.pops = []
            scenario.reset_world(world)
        return list(rec.pops)
    try:
        return sizes(_Recorder)
    except NeedConcretePick:
        raise
    except Exception:
        return sizes(_HybridRecorder)          # (a host reset: its numbers drawn for real)




def topo(roots):
    """Nodes reachable from `roots`, arguments before users (trace order where it matters: uid order is creation order)."""
    seen, order, stack = set(), [], [(r, False) for r in reversed(list(roots))]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n)
            continue
        if n.uid in seen:
            continue
        seen.add(n.uid)
        stack.append((n, True))
        for a in reversed(n.args):
            if a.uid not in seen:
                stack.append((a, False))
    return order


def _topo_stop(roots, stop):
    """topo(roots) that does not descend below the nodes in `stop` (they are inputs there)"""
    seen, order, stack = set(), [], [(r, False) for r in reversed(list(roots))]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n)
            continue
        if n.uid in seen:
            continue
        seen.add(n.uid)
        stack.append((n, True))
        if n.uid in stop:
            continue
        for a in reversed(n.args):
            if a.uid not in seen:
                stack.append((a, False))
    return order


def inputs_of(roots):
    return set(n.op for n in topo(roots) if n.op in ("P", "V", "C", "K", "U"))


# ---- evaluation with NumPy (fp64 by default), vectorised over worlds: the verification, the CPU tests, host-side resets ------
def decision_margin(roots, B, **state):
    """Per world: how far the nearest of the graph's comparisons is from flipping -- min over its lt / le / eq / ne nodes of
    |lhs - rhs| (fp64; the ordering tests `<` / `<=` -- equality tests compare exact data).  Outputs of an fp32 evaluation can legitimately differ from the fp64 one where this is ~1e-6: tests
    compare outside such a band (as the contact-count tests of the fused kernels do)."""
    margin = np.full(B, np.inf)
    evaluate(roots, B, _margin=margin, **state)
    return margin


def evaluate(roots, B, P=None, V=None, Cw=None, K=None, U=None, dtype=np.float64, _margin=None):
    """Values of `roots` for B worlds: P [B, E, 2], V [B, E, 2], Cw [B, A, dim_c], K [B, n_picks] (ints), U [B, n_draws]."""
    val = {}
    one = np.ones(B, dtype)
    with np.errstate(all="ignore"):
        for n in topo(roots):
            a = [val[x.uid] for x in n.args]
            op = n.op
            if op == "const":
                v = one * dtype(n.value)
            elif op == "bconst":
                v = np.full(B, n.value, bool)
            elif op == "P":
                v = P[:, n.value[0], n.value[1]].astype(dtype)
            elif op == "V":
                v = V[:, n.value[0], n.value[1]].astype(dtype)
            elif op == "C":
                v = Cw[:, n.value[0], n.value[1]].astype(dtype)
            elif op == "K":
                v = K[:, n.value[0]].astype(dtype)
            elif op == "U":
                v = U[:, n.value[0]].astype(dtype)
            elif op == "add":
                v = a[0] + a[1]
            elif op == "sub":
                v = a[0] - a[1]
            elif op == "mul":
                v = a[0] * a[1]
            elif op == "div":
                v = a[0] / a[1]
            elif op == "min":
                v = np.where(a[1] < a[0], a[1], a[0])      # Python's min(a, b): b only if b < a
            elif op == "max":
                v = np.where(a[1] > a[0], a[1], a[0])
            elif op == "neg":
                v = -a[0]
            elif op == "abs":
                v = np.abs(a[0])
            elif op == "sqrt":
                v = np.sqrt(a[0])
            elif op == "exp":
                v = np.exp(a[0])
            elif op == "log":
                v = np.log(a[0])
            elif op == "tanh":
                v = np.tanh(a[0])
            elif op == "sin":
                v = np.sin(a[0])
            elif op == "cos":
                v = np.cos(a[0])
            elif op == "atan2":
                v = np.arctan2(a[0], a[1])
            elif op == "pow":
                v = np.power(a[0], a[1])
            elif op == "mod":
                v = np.mod(a[0], a[1])
                if _margin is not None:      # (the wrap: where a / b passes an integer)
                    q = a[0] / a[1]
                    f = q - np.floor(q)
                    d = np.minimum(f, 1.0 - f) * np.abs(a[1])
                    np.minimum(_margin, np.where(np.isfinite(d), d, np.inf), out=_margin)
            elif op == "f32":
                v = a[0].astype(np.float32).astype(a[0].dtype)
            elif op in ("floor", "rint"):
                v = np.floor(a[0]) if op == "floor" else np.rint(a[0])
                if _margin is not None:      # how far the argument is from the next step of the staircase
                    f = a[0] - np.floor(a[0])
                    d = np.minimum(f, 1.0 - f) if op == "floor" else np.abs(f - 0.5)
                    np.minimum(_margin, np.where(np.isfinite(d), d, np.inf), out=_margin)
            elif op in ("lt", "le") and _margin is not None and not any(x.op == "K" or x.op == "sel" for x in n.args):
                # (== / != are tests on exact data -- an utterance that is all zeros, simple_crypto.py:104 -- and hold in fp32 as in fp64)
                np.minimum(_margin, np.where(np.isfinite(a[0] - a[1]), np.abs(a[0] - a[1]), np.inf), out=_margin)
                v = {"lt": a[0] < a[1], "le": a[0] <= a[1], "eq": a[0] == a[1], "ne": a[0] != a[1]}[op]
            elif op == "lt":
                v = a[0] < a[1]
            elif op == "le":
                v = a[0] <= a[1]
            elif op == "eq":
                v = a[0] == a[1]
            elif op == "ne":
                v = a[0] != a[1]
            elif op == "not":
                v = ~a[0]
            elif op == "and":
                v = a[0] & a[1]
            elif op == "or":
                v = a[0] | a[1]
            elif op == "ite":
                v = np.where(a[0], a[1], a[2])
            elif op == "sel":
                k = a[0].astype(np.int64)
                v = np.choose(np.clip(k, 0, len(a) - 2), a[1:])
            else:
                raise ValueError("node %r" % op)
            val[n.uid] = v
    return [val[r.uid] for r in roots]


# ---- verification: the graphs against the file's own callbacks, run concretely -------------------------------------------------
def _concrete_reset(scenario, world, u, k):
    with patched_random(_Replayer(u, k)):
        scenario.reset_world(world)


def random_states(t, R, rs, spread=1.0):
    """R random worlds' post-step states: clustered enough that contacts, overlaps and arena exits all occur."""
    scale = rs.choice([0.15, 0.4, 1.0, 1.3], size=(R, 1, 1)) * spread
    P = rs.uniform(-1.0, 1.0, (R, t.E, 2)) * scale
    V = rs.uniform(-1.3, 1.3, (R, t.E, 2))
    V[:, t.A:] = 0.0
    Cw = np.zeros((R, t.A, max(t.dim_c, 1)))
    if t.dim_c:
        words = rs.randint(0, t.dim_c + 1, (R, t.A))           # == dim_c: says nothing (zeros)
        for c in range(t.dim_c):
            Cw[:, :, c] = (words == c) * rs.choice([1.0, 0.5], size=(R, t.A))
    return P, V, Cw[:, :, :t.dim_c] if t.dim_c else Cw[:, :, :0]


def verify(scenario, t, worlds=96, seed=0, tol=None):
    """Evaluate the trace with NumPy (fp64) on `worlds` random worlds and compare with the file's own reset_world / observation /
    reward / done run concretely on the same worlds.  Returns the largest scaled difference; raises TraceUnsupported when the
    trace does not reproduce the file (hidden state, randomness, something the tracer mis-models)."""
    rs = np.random.RandomState(seed)
    R = int(worlds)
    if tol is None:
        # a file that computes in float32 (np.float32(x) * y, arrays of dtype float32) rounds after every operation; the graph rounds
        # where the file converts (f32 nodes) and computes in double in between: single-precision noise, not a modelling error
        every = [n for row in t.obs for n in row] + list(t.rew) + [n for row in (getattr(t, "info", None) or []) for n in row]
        tol = 2e-6 if any(n.op == "f32" for n in topo(every)) else 1e-9
    rng_state = np.random.get_state()
    try:
        cw = scenario.make_world()
    finally:
        np.random.set_state(rng_state)
    agents, ents = _entity_lists(cw)
    U = rs.uniform(0.0, 1.0, (R, max(t.n_u, 1)))
    K = np.stack([rs.randint(0, n, R) for n in t.pops], axis=1) if t.pops else np.zeros((R, 0), np.int64)
    for j, u in enumerate(getattr(t, "params", ())):          # the draws the callbacks read: the value their pick slot carries
        U[:, u] = K[:, t.real_picks() + j] / float(PARAM_POP)
    P, V, Cw = random_states(t, R, rs)
    worst = 0.0
    if getattr(t, "host_reset", None):          # the picks of a host reset are whatever its own run draws: one run per world to learn them
        host_seeds = rs.randint(0, 2 ** 31 - 1, R)
        rng_state = np.random.get_state()
        try:
            for r in range(R):
                np.random.seed(int(host_seeds[r]))
                with logged_picks(PickLogger()) as lg:
                    scenario.reset_world(cw)
                if len(lg.log) != t.real_picks():
                    raise TraceUnsupported("reset_world makes %d picks in one world and %d in another" % (t.real_picks(), len(lg.log)))
                K[r, :t.real_picks()] = lg.log
        finally:
            np.random.set_state(rng_state)

    def cmp(what, got, want):
        nonlocal worst
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        both_nan = np.isnan(got) & np.isnan(want)
        err = np.where(both_nan, 0.0, np.abs(got - want) / np.maximum(1.0, np.abs(want)))
        if not np.all(err <= tol):
            raise TraceUnsupported("the trace does not reproduce the file's %s (scaled difference %.3e): hidden state, randomness, "
                                   "or an operation the tracer does not model" % (what, float(np.nanmax(err))))
        worst = max(worst, float(err.max()) if err.size else 0.0)
    flat_pos = [n for e in t.reset_pos for n in e]
    flat_vel = [n for e in t.reset_vel for n in e]
    flat_c = [n for a in t.reset_c for n in a]
    rp = np.stack(evaluate(flat_pos, R, K=K, U=U), axis=1).reshape(R, t.E, 2)
    rv = np.stack(evaluate(flat_vel, R, K=K, U=U), axis=1).reshape(R, t.E, 2)
    rc = np.stack(evaluate(flat_c, R, K=K, U=U), axis=1).reshape(R, t.A, t.dim_c) if flat_c else np.zeros((R, t.A, 0))
    roots = [n for row in t.obs for n in row] + list(t.rew) + [d for d in t.done if d is not None]
    vals = evaluate(roots, R, P=P, V=V, Cw=Cw, K=K, U=U)
    widths = [len(row) for row in t.obs]
    off = np.cumsum([0] + widths)
    obs_eval = [np.stack(vals[off[i]:off[i + 1]], axis=1) if widths[i] else np.zeros((R, 0)) for i in range(t.A)]
    rew_eval = vals[off[-1]:off[-1] + t.A]
    done_eval, q = [], off[-1] + t.A
    for d in t.done:
        done_eval.append(vals[q] if d is not None else None)
        q += d is not None
    info_eval = None
    if getattr(t, "info", None) is not None:
        info_eval = [evaluate(row, R, P=P, V=V, Cw=Cw, K=K, U=U) if row else [] for row in t.info]
    for r in range(R):
        if getattr(t, "host_reset", None):      # the file's own reset_world, run as it is with world r's picks (what else it hides shows below)
            rng_state = np.random.get_state()
            try:
                np.random.seed(int(host_seeds[r]))
                with logged_picks(PickLogger(forced=K[r])):
                    scenario.reset_world(cw)
            finally:
                np.random.set_state(rng_state)
        else:
            _concrete_reset(scenario, cw, U[r], K[r])
            cmp("reset_world positions", [e.state.p_pos for e in ents], rp[r])
            cmp("reset_world velocities", [np.zeros(2) if e.state.p_vel is None else e.state.p_vel for e in ents], rv[r])
            if t.dim_c:
                cmp("reset_world utterances", [np.zeros(t.dim_c) if a.state.c is None else a.state.c for a in agents], rc[r])
        for k, e in enumerate(ents):
            e.state.p_pos = P[r, k].copy()
            e.state.p_vel = V[r, k].copy()
        for i, a in enumerate(agents):
            a.state.c = np.zeros(t.dim_c) if a.silent else Cw[r, i].copy()
        for i, a in enumerate(agents):
            o = np.asarray(scenario.observation(a, cw), np.float64).reshape(-1)
            if o.shape[0] != widths[i]:
                raise TraceUnsupported("observation of agent %d has %d columns concretely, %d traced" % (i, o.shape[0], widths[i]))
            cmp("observation (agent %d)" % i, o, obs_eval[i][r])
            cmp("reward (agent %d)" % i, float(scenario.reward(a, cw)), rew_eval[i][r])
            if t.done[i] is not None:
                if bool(scenario.done(a, cw)) != bool(done_eval[i][r]):
                    raise TraceUnsupported("the trace does not reproduce the file's done (agent %d)" % i)
            if getattr(t, "info", None) is not None:
                raw = scenario.benchmark_data(a, cw)
                if _describe_info(raw) != t.info_desc[i]:
                    raise TraceUnsupported("benchmark_data of agent %d changes its structure from world to world" % i)
                want = [float(x) for x in _concrete_info(raw, t.info_desc[i])]
                cmp("benchmark_data (agent %d)" % i, want, [info_eval[i][k][r] for k in range(len(want))])
    return worst


def _concrete_info(raw, desc):
    if desc[0] == "none":
        return []
    if desc[0] == "s":
        return [np.asarray(raw).reshape(-1)[0]]
    if desc[0] == "a":
        return list(np.asarray(raw).reshape(-1))
    return [v for x, d in zip(raw, desc[1]) for v in _concrete_info(x, d)]


def evaluate_torch(roots, B, K=None, U=None, P=None, V=None, Cw=None, device=None):
    """`evaluate` with torch ops on `device` (fp32): a traced reset_world that is not World.reset_uniform's placement is drawn
    and evaluated for all worlds at once without leaving the device.  K [n_picks, B] int64, U [n_draws, B]; -> list of [B] tensors."""
    import torch
    val = {}

    def full(x, dtype=torch.float32):
        return torch.full((B,), x, dtype=dtype, device=device)
    for n in topo(roots):
        a = [val[x.uid] for x in n.args]
        op = n.op
        if op == "const":
            v = full(float(n.value))
        elif op == "bconst":
            v = full(bool(n.value), torch.bool)
        elif op == "U":
            v = U[n.value[0]]
        elif op == "K":
            v = K[n.value[0]].to(torch.float32)
        elif op == "P":
            v = P[n.value[0], n.value[1]]
        elif op == "V":
            v = V[n.value[0], n.value[1]]
        elif op == "C":
            v = Cw[n.value[0]][:, n.value[1]]
        elif op in ("add", "sub", "mul", "div"):
            v = {"add": torch.add, "sub": torch.sub, "mul": torch.mul, "div": torch.div}[op](a[0], a[1])
        elif op == "min":
            v = torch.where(a[1] < a[0], a[1], a[0])
        elif op == "max":
            v = torch.where(a[1] > a[0], a[1], a[0])
        elif op in ("neg", "abs", "sqrt", "exp", "log", "tanh", "sin", "cos"):
            v = getattr(torch, op)(a[0])
        elif op == "atan2":
            v = torch.atan2(a[0], a[1])
        elif op == "pow":
            v = torch.pow(a[0], a[1])
        elif op == "mod":
            v = torch.remainder(a[0], a[1])
        elif op == "f32":
            v = a[0]
        elif op == "floor":
            v = torch.floor(a[0])
        elif op == "rint":
            v = torch.round(a[0])
        elif op in _CMP:
            v = {"lt": torch.lt, "le": torch.le, "eq": torch.eq, "ne": torch.ne}[op](a[0], a[1])
        elif op == "not":
            v = ~a[0]
        elif op == "and":
            v = a[0] & a[1]
        elif op == "or":
            v = a[0] | a[1]
        elif op == "ite":
            v = torch.where(a[0], a[1], a[2])
        elif op == "sel":
            k = a[0].to(torch.int64).clamp(0, len(a) - 2)
            v = torch.stack([x.to(torch.float32) for x in a[1:]], dim=0).gather(0, k[None, :])[0]
        else:
            raise ValueError("node %r" % op)
        val[n.uid] = v
    return [val[r.uid] for r in roots]


# ---- device code: one statement per node, in trace order ------------------------------------------------------------------------
def _f32_literal(v):
    v = float(np.float32(v))
    if v != v:
        return "__builtin_nanf(\"\")"
    if v in (float("inf"), float("-inf")):
        return ("-" if v < 0 else "") + "__builtin_inff()"
    return "%sf" % float(v).hex()          # C++17 hexadecimal floating literal: the fp32 value, exactly


def _emit(roots, lines, names, shared=None):
    """Append statements computing `roots` to `lines`; `names`: uid -> expression (variable name or literal) so far; `shared`:
    uid -> slot of the values traced_shared has computed (read through S instead of being recomputed)."""
    def ref(n):
        return names[n.uid]
    stop = set(shared or ())
    for n in _topo_stop(roots, stop):
        if n.uid in names:
            continue
        op, a = n.op, n.args
        if n.uid in stop:
            lines.append("      const float t%d = S(%d);" % (n.uid, shared[n.uid]))
            names[n.uid] = "t%d" % n.uid
            continue
        var = ("b%d" if n.is_bool else "t%d") % n.uid
        if op == "const":
            names[n.uid] = _f32_literal(n.value)
            continue
        if op == "bconst":
            names[n.uid] = "true" if n.value else "false"
            continue
        if op == "P":
            e = "P(%d, %d)" % n.value
        elif op == "V":
            e = "V(%d, %d)" % n.value
        elif op == "C":
            e = "W(%d, %d)" % n.value
        elif op == "K":
            lines.append("      const int k%d = K(%d);" % (n.uid, n.value[0]))
            e = "(float)k%d" % n.uid
        elif op == "U":
            raise TraceUnsupported("a uniform draw in device code")
        elif op in ("add", "sub", "mul", "div"):
            e = "%s %s %s" % (ref(a[0]), {"add": "+", "sub": "-", "mul": "*", "div": "/"}[op], ref(a[1]))
        elif op == "min":
            e = "(%s < %s ? %s : %s)" % (ref(a[1]), ref(a[0]), ref(a[1]), ref(a[0]))        # Python's min(a, b): b only if b < a
        elif op == "max":
            e = "(%s > %s ? %s : %s)" % (ref(a[1]), ref(a[0]), ref(a[1]), ref(a[0]))
        elif op == "neg":
            e = "-%s" % ref(a[0])
        elif op == "abs":
            e = "fabsf(%s)" % ref(a[0])
        elif op == "sqrt":
            e = "fast_sqrt(%s)" % ref(a[0])
        elif op == "exp":
            e = "__builtin_amdgcn_exp2f(%s * 1.44269504088896341f)" % ref(a[0])
        elif op == "log":
            e = "(__builtin_amdgcn_logf(%s) * 0.693147180559945309f)" % ref(a[0])
        elif op in ("tanh", "sin", "cos"):
            e = "%sf(%s)" % (op, ref(a[0]))
        elif op == "atan2":
            e = "atan2f(%s, %s)" % (ref(a[0]), ref(a[1]))
        elif op == "pow":
            e = "powf(%s, %s)" % (ref(a[0]), ref(a[1]))
        elif op == "f32":          # (the device computes in single precision anyway)
            names[n.uid] = ref(a[0])
            continue
        elif op in ("floor", "rint"):
            e = "%sf(%s)" % (op, ref(a[0]))
        elif op == "mod":      # NumPy's / Python's %: fmod, moved to the divisor's sign
            lines.append("      const float m%d = fmodf(%s, %s);" % (n.uid, ref(a[0]), ref(a[1])))
            e = "(m%d != 0.0f && ((m%d < 0.0f) != (%s < 0.0f))) ? m%d + %s : m%d" % (n.uid, n.uid, ref(a[1]), n.uid, ref(a[1]), n.uid)
        elif op in ("lt", "le"):
            # `np.sqrt(np.sum(np.square(d))) < r`, the reference's contact test (simple_tag.py:69-73): decided as NumPy's float32
            # rounding sequence would (sqrt_lt: exact, without the correctly rounded sqrt outside a 1e-6 band around r)
            # (a square root that traced_shared computed is read through S: its argument does not exist here)
            own = [x.op == "sqrt" and x.uid not in stop for x in a]
            if op == "lt" and own[0]:
                e = "sqrt_lt(%s, %s)" % (ref(a[0].args[0]), ref(a[1]))
            elif own[0] or own[1]:
                l = "sqrtf(%s)" % ref(a[0].args[0]) if own[0] else ref(a[0])
                r = "sqrtf(%s)" % ref(a[1].args[0]) if own[1] else ref(a[1])
                e = "%s %s %s" % (l, "<" if op == "lt" else "<=", r)
            else:
                e = "%s %s %s" % (ref(a[0]), "<" if op == "lt" else "<=", ref(a[1]))
        elif op in ("eq", "ne"):
            e = "%s %s %s" % (ref(a[0]), "==" if op == "eq" else "!=", ref(a[1]))
        elif op == "not":
            e = "!%s" % ref(a[0])
        elif op in ("and", "or"):
            e = "%s %s %s" % (ref(a[0]), "&&" if op == "and" else "||", ref(a[1]))
        elif op == "ite":
            e = "%s ? %s : %s" % (ref(a[0]), ref(a[1]), ref(a[2]))
        elif op == "sel":
            kn = a[0]
            kv = "k%d" % kn.uid if kn.op == "K" else "(int)%s" % ref(kn)
            e = ref(a[-1])
            for j in range(len(a) - 2, 0, -1):
                e = "%s == %d ? %s : (%s)" % (kv, j - 1, ref(a[j]), e)
        else:
            raise TraceUnsupported("no device code for node %r" % op)
        lines.append("      const %s %s = %s;" % ("bool" if n.is_bool else "float", var, e))
        names[n.uid] = var
    return [names[r.uid] for r in roots]


def _device_form(g, roots):
    """The graphs as they are GENERATED (what is verified and evaluated on the host stays the trace itself): the smallest / largest
    of several distances as ONE square root -- min(sqrt(a), sqrt(b)) == sqrt(min(a, b)), the form the hand-written kernels use
    (`fast_sqrt(min_d2_run(...))`): a reward that takes, per landmark, the distance of the nearest of N agents evaluates N square
    roots (quarter-rate instructions) instead of N^2."""
    memo = {}

    def rebuild(n):
        r = memo.get(n.uid)
        if r is not None:
            return r
        args = tuple(rebuild(a) for a in n.args)
        if n.op in ("min", "max") and args[0].op == "sqrt" and args[1].op == "sqrt":
            r = g.unary("sqrt", g.binary(n.op, args[0].args[0], args[1].args[0]))
        elif all(a is b for a, b in zip(args, n.args)):
            r = n
        else:
            r = g.node(n.op, args, n.value)
        memo[n.uid] = r
        return r
    # (iteratively deep graphs: accumulation chains of a few hundred nodes -- within Python's recursion limit for the program sizes
    #  hip_source accepts; topo order makes every argument available before its user)
    for n in topo(roots):
        rebuild(n)
    return [memo[r.uid] for r in roots]


_SHARE_MIN_COST = 8          # a value worth a trip through LDS
_SHARE_MIN_SAVING = 300      # statements per world saved, below which the extra phase (tasks, a barrier) is not worth having
_SHARE_MAX = 64


def shared_tasks(rew_roots, n_waves):
    """Which parts of the agents' reward graphs to compute ONCE per world (traced_shared) instead of once per agent: the reference's
    cooperative rewards are the same expression for every agent up to a few agent-specific terms (simple_spread.py:72-82: the
    distance of the nearest agent to every landmark, then the agent's own collisions), and each agent's wave would evaluate all of
    it.  Top-down from every reward root: a node used by two agents or more whose cone is not trivial becomes a task if it is
    small enough to leave work for every wave (else its arguments are looked at, and the node itself is recomputed by each agent
    from their values); nothing is shared unless it saves a few hundred statements per world.  -> list of task nodes."""
    roots = list(rew_roots)
    if len(roots) < 2:
        return []
    users = {}
    for i, r in enumerate(roots):
        for n in topo([r]):
            users.setdefault(n.uid, set()).add(i)
    leaves = ("const", "bconst", "P", "V", "C", "K", "U")
    memo = {}

    class _Cost(object):          # statements a node stands for: the operations in its cone, each counted once
        def __getitem__(self, uid):
            return memo[uid]

    def cone(n):
        c = memo.get(n.uid)
        if c is None:
            c = memo[n.uid] = sum(1 for x in topo([n]) if x.op not in leaves)
        return c
    cost = _Cost()
    per_agent = max(cone(r) for r in roots)
    c_max = max(32, per_agent // max(1, n_waves))
    while True:
        tasks, seen = [], set()

        def descend(n):
            if n.uid in seen:
                return
            seen.add(n.uid)
            if len(users[n.uid]) >= 2 and not n.is_bool and n.op not in leaves and _SHARE_MIN_COST <= cone(n) <= c_max:
                tasks.append(n)
                return
            for a in n.args:
                descend(a)
        import sys
        limit = sys.getrecursionlimit()
        sys.setrecursionlimit(max(limit, 20000))
        try:
            for r in roots:
                descend(r)
        finally:
            sys.setrecursionlimit(limit)
        if len(tasks) <= _SHARE_MAX:
            break
        if c_max >= per_agent:
            # no cone is larger than per_agent: a bigger limit cannot merge anything further (more than _SHARE_MAX distinct
            # shareable terms under agent-specific chains).  Share the terms that save most; the others stay in every
            # agent's own code, which is always correct.
            tasks.sort(key=lambda n: (-cone(n) * (len(users[n.uid]) - 1), n.uid))
            tasks = sorted(tasks[:_SHARE_MAX], key=lambda n: n.uid)
            break
        c_max *= 2
    saving = sum(cost[n.uid] * (len(users[n.uid]) - 1) for n in tasks)
    return tasks if saving >= _SHARE_MIN_SAVING else []


def hip_source(t, n_waves=None):
    """The device functions of a trace: what is appended to the generated header of the compiled row program (the kernel calls
    them through the ops MPE_ROW_OBS_CODE / R_CODE / R_DONE_CODE).  P, V, W, K are the kernel's accessors of the staged state."""
    out = ["", "// ---- traced callbacks (symtrace.py): one statement per arithmetic step of the file's NumPy code, in its order ----",
           "#define MPE_ROWS_TRACED 1", '#include "mpe_internal.h"', "namespace mpe {", "namespace {",
           "template <class FP, class FV, class FW, class FK>",
           "__device__ __forceinline__ void traced_obs(const int i, float *const row, const FP &P, const FV &V, const FW &W, const FK &K) {",
           "  switch (i) {"]
    g = getattr(t, "graph", None)
    form = (lambda roots: _device_form(g, roots)) if g is not None else (lambda roots: list(roots))
    rew_roots = form(list(t.rew))
    tasks = shared_tasks(rew_roots, n_waves if n_waves else min(len(t.rew), 16)) if getattr(t, "share", True) else []
    shared = {n.uid: k for k, n in enumerate(tasks)}
    t.n_shared = len(tasks)
    for i, row in enumerate(t.obs):
        lines, names = [], {}
        vals = _emit(form(row), lines, names)
        out.append("    case %d: {" % i)
        out += lines
        out += ["      row[%d] = %s;" % (j, v) for j, v in enumerate(vals)]
        out.append("    } break;")
    out += ["    default: break;", "  }", "}",
            "// what several agents' rewards share, once per world: task k's value into its LDS slot (read back through S(k))",
            "template <class FP, class FV, class FW, class FK>",
            "__device__ __forceinline__ void traced_shared(const int k, float *const out, const FP &P, const FV &V, const FW &W, const FK &K) {",
            "  switch (k) {"]
    for k, n in enumerate(tasks):
        lines, names = [], {}
        v = _emit([n], lines, names)[0]
        out.append("    case %d: {" % k)
        out += lines
        out.append("      out[0] = %s;" % v)
        out.append("    } break;")
    out += ["    default: break;", "  }", "}",
            "template <class FP, class FV, class FW, class FK, class FS>",
            "__device__ __forceinline__ float traced_rew(const int i, const FP &P, const FV &V, const FW &W, const FK &K, const FS &S) {",
            "  switch (i) {"]
    for i, r in enumerate(rew_roots):
        lines, names = [], {}
        v = _emit([r], lines, names, shared)[0]
        out.append("    case %d: {" % i)
        out += lines
        out.append("      return %s;" % v)
        out.append("    }")
    out += ["    default: return 0.f;", "  }", "}",
            "template <class FP, class FV, class FW, class FK>",
            "__device__ __forceinline__ bool traced_done(const int i, const FP &P, const FV &V, const FW &W, const FK &K) {",
            "  switch (i) {"]
    for i, d in enumerate(t.done):
        if d is None:
            continue
        lines, names = [], {}
        v = _emit([d], lines, names)[0]
        out.append("    case %d: {" % i)
        out += lines
        out.append("      return %s;" % v)
        out.append("    }")
    out += ["    default: return false;", "  }", "}", "}  // namespace", "}  // namespace mpe", ""]
    # straight-line code: every agent's functions spell out their whole graph (a reward that visits every agent-landmark pair is
    # N^2 statements per agent, N^3 per program) -- past a size the compiler takes minutes (N = 16 cooperative navigation: 35 K
    # statements, 2 minutes of hipcc, a 1.6 MB image).  Larger programs are refused: the file runs on the host path, and teams of
    # that size are what the built-in kernels / looped row programs are for.
    import os
    limit = int(os.environ.get("MPE_TRACE_MAX_STATEMENTS", "60000"))
    if len(out) > limit:
        raise TraceUnsupported("the traced program is %d statements of straight-line device code (limit %d, MPE_TRACE_MAX_STATEMENTS): "
                               "too large a team for per-agent generated code" % (len(out), limit))
    return "\n".join(out)
