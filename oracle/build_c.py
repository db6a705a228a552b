"""Builds oracle/mpe_oracle.c (the plain-C restatement; TEST INFRASTRUCTURE, see oracle/__init__.py) into
oracle/_build/libmpe_oracle.so with gcc, and binds it with ctypes.  Called by __graft_entry__.build()
("building the checker is not using it"), by tests/test_oracle_golden.py and by bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mpe_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libmpe_oracle.so")
MAX_E = 256
KIND = {"simple": 1, "simple_spread": 2, "simple_tag": 3}


class OrcSpec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_agents", C.c_int32), ("n_landmarks", C.c_int32), ("dim_c", C.c_int32),
                ("n_adversaries", C.c_int32), ("collaborative", C.c_int32),
                ("dt", C.c_double), ("damping", C.c_double), ("contact_force", C.c_double), ("contact_margin", C.c_double),
                ("size", C.c_double * MAX_E), ("mass", C.c_double * MAX_E), ("accel", C.c_double * MAX_E),
                ("max_speed", C.c_double * MAX_E), ("movable", C.c_uint8 * MAX_E), ("collide", C.c_uint8 * MAX_E)]


def build(verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-o", LIB, SRC, "-lm"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed:\n" + r.stderr[-4000:])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        h = C.CDLL(build(verbose=False))
        h.orc_sizeof_spec.restype = C.c_size_t
        assert h.orc_sizeof_spec() == C.sizeof(OrcSpec), "OrcSpec layout mismatch"
        h.orc_obs_total.argtypes = [C.POINTER(OrcSpec)]
        h.orc_step_batch.argtypes = [C.POINTER(OrcSpec), C.c_int64] + [C.c_void_p] * 6 + [C.c_int]
        h.orc_bench.restype = C.c_double
        h.orc_bench.argtypes = [C.POINTER(OrcSpec), C.c_double, C.c_int, C.c_int, C.c_double]
        _lib = h
    return _lib


def c_spec(spec):
    """oracle.spec.Spec -> OrcSpec."""
    s = OrcSpec()
    s.kind, s.n_agents, s.n_landmarks, s.dim_c = KIND[spec.name], spec.n_agents, spec.n_landmarks, spec.dim_c
    s.n_adversaries = sum(1 for a in spec.adversary if a)
    s.collaborative = 1 if spec.collaborative else 0
    s.dt, s.damping, s.contact_force, s.contact_margin = spec.dt, spec.damping, spec.contact_force, spec.contact_margin
    for e in range(spec.n_entities):
        s.size[e], s.mass[e] = spec.size[e], spec.mass_of(e)
        s.movable[e], s.collide[e] = int(spec.movable[e]), int(spec.collide[e])
    for i in range(spec.n_agents):
        s.accel[i] = 5.0 if spec.accel[i] is None else spec.accel[i]
        s.max_speed[i] = -1.0 if spec.max_speed[i] is None else spec.max_speed[i]
    return s


def step_batch(spec, pos, vel, act, threads=1):
    """pos [B,E,2], vel [B,A,2], act [B,A,5] (fp64) -> (pos, vel, obs list of [B,D_i], rew [B,A], collisions [B,A])."""
    import numpy as np
    s = c_spec(spec)
    L = lib()
    B, A = pos.shape[0], spec.n_agents
    pos = np.ascontiguousarray(pos, np.float64).copy()
    vel = np.ascontiguousarray(vel, np.float64).copy()
    act = np.ascontiguousarray(act, np.float64)
    D = L.orc_obs_total(C.byref(s))
    obs = np.zeros((B, D), np.float64)
    rew = np.zeros((B, A), np.float64)
    col = np.zeros((B, A), np.int32)
    L.orc_step_batch(C.byref(s), B, pos.ctypes.data, vel.ctypes.data, act.ctypes.data, obs.ctypes.data,
                     rew.ctypes.data, col.ctypes.data, threads)
    dims = spec.obs_dims()
    offs = np.concatenate([[0], np.cumsum(dims)])
    return pos, vel, [obs[:, offs[i]:offs[i + 1]] for i in range(A)], rew, col


def bench(spec, seconds, threads, episode_len=25):
    return float(lib().orc_bench(C.byref(c_spec(spec)), seconds, threads, episode_len, spec.landmark_range))


if __name__ == "__main__":
    print(build())
