"""ctypes binding of libmpe_hip.so (the C ABI declared in include/mpe_hip.h).

There is no CPU fallback: if the shared object is missing this module raises at import of the
library handle (`lib()`), and every compute entry point needs device pointers.
"""
import ctypes as C
import os

MPE_ABI_VERSION = 4
MPE_MAX_ENTITIES = 512
MPE_ACTION_DIM = 5
MPE_SCN_GENERIC, MPE_SCN_SIMPLE, MPE_SCN_SPREAD, MPE_SCN_TAG, MPE_SCN_ADVERSARY, MPE_SCN_PUSH = 0, 1, 2, 3, 4, 5
MPE_SCN_SPEAKER_LISTENER, MPE_SCN_REFERENCE, MPE_SCN_CRYPTO, MPE_SCN_WORLD_COMM = 6, 7, 8, 9
COMM_KINDS = (MPE_SCN_SPEAKER_LISTENER, MPE_SCN_REFERENCE, MPE_SCN_CRYPTO, MPE_SCN_WORLD_COMM)   # agents speak: MpeBuffers.comm
MPE_MAX_CHOICES = 4
# the composable output stage (include/mpe_hip.h, enum MpeRowOp)
MPE_ROWS_MAX_ENTITIES = 64
MPE_ROW_SELF = 255
MPE_ROWS_HEADER_BYTES = 4096
(MPE_ROW_OBS_VEL, MPE_ROW_OBS_POS, MPE_ROW_OBS_REL, MPE_ROW_OBS_REL_PICK, MPE_ROW_OBS_COMM, MPE_ROW_OBS_CONST, MPE_ROW_OBS_ONEHOT,
 MPE_ROW_OBS_REL_VIS, MPE_ROW_OBS_VEL_VIS, MPE_ROW_OBS_IN_REGION) = range(1, 11)
(MPE_ROW_OBS_REL_RANGE, MPE_ROW_OBS_VEL_RANGE, MPE_ROW_OBS_REL_VIS_RANGE, MPE_ROW_OBS_VEL_VIS_RANGE, MPE_ROW_OBS_CONST_N) = range(11, 16)
MPE_ROW_R_MIN_D2_RANGE, MPE_ROW_R_MIN_D2_TO_RANGE, MPE_ROW_R_ADD_IF_HIT_GRID, MPE_ROW_R_ADD_MIN_DIST_GRID = 48, 49, 50, 51
MPE_ROW_R_ABS_POS, MPE_ROW_R_DONE_IF_GT, MPE_ROW_R_DONE_IF_LT, MPE_ROW_R_DONE_IF_HIT = 52, 53, 54, 55
MPE_ROW_OBS_CODE, MPE_ROW_R_CODE, MPE_ROW_R_DONE_CODE = 16, 56, 57      # traced code (symtrace.py): compiled-in programs only
(MPE_ROW_R_D2, MPE_ROW_R_MIN_D2, MPE_ROW_R_D2_PICK, MPE_ROW_R_MIN_D2_PICK, MPE_ROW_R_SQRT, MPE_ROW_R_BOUND, MPE_ROW_R_COMM_ERR,
 MPE_ROW_R_COMM_SUM, MPE_ROW_R_CONST, MPE_ROW_R_SAVE, MPE_ROW_R_LOAD, MPE_ROW_R_ZERO, MPE_ROW_R_ADD, MPE_ROW_R_ADD_IF_HIT,
 MPE_ROW_R_ADD_ACC, MPE_ROW_R_STORE) = range(32, 48)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPE_HIP_LIB") or os.path.join(_HERE, "lib", "libmpe_hip.so")  # override: A/B builds

_M = MPE_MAX_ENTITIES


class MpeScenarioDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n_agents", C.c_int32), ("n_landmarks", C.c_int32), ("dim_c", C.c_int32),
        ("n_adversaries", C.c_int32), ("collaborative", C.c_int32),
        ("dt", C.c_float), ("damping", C.c_float), ("contact_force", C.c_float), ("contact_margin", C.c_float),
        ("size", C.c_float * _M), ("mass", C.c_float * _M), ("accel", C.c_float * _M),
        ("max_speed", C.c_float * _M), ("movable", C.c_uint8 * _M), ("collide", C.c_uint8 * _M),
        ("obs_off", C.c_int32 * (_M + 1)),
        ("n_choices", C.c_int32), ("choice_pop", C.c_int32 * MPE_MAX_CHOICES),
    ]


class MpeStepServer(C.Structure):      # include/mpe_hip.h: the step server's device words and ring geometry
    _fields_ = [("door", C.c_void_p), ("flag", C.c_void_p), ("status", C.c_void_p), ("act_ring", C.c_void_p),
                ("ring", C.c_int32), ("slots", C.c_int32), ("timeout_us", C.c_uint64), ("comm_ring", C.c_void_p),
                ("ahead", C.c_int32), ("reserved_", C.c_int32)]


class MpeBuffers(C.Structure):
    _fields_ = [
        ("pos", C.c_void_p), ("vel", C.c_void_p), ("act", C.c_void_p), ("ids", C.c_void_p), ("u", C.c_void_p),
        ("obs", C.c_void_p), ("rew", C.c_void_p), ("done", C.c_void_p),
        ("info_rew", C.c_void_p), ("info_collisions", C.c_void_p), ("info_min_dists", C.c_void_p),
        ("info_occupied", C.c_void_p), ("force", C.c_void_p), ("entity_table", C.c_void_p),
        ("comm", C.c_void_p), ("choice", C.c_void_p),
    ]


class MpeRowProgram(C.Structure):
    _fields_ = [
        ("ops_device", C.c_void_p), ("header_device", C.c_void_p), ("header_hash", C.c_uint64), ("n_ops", C.c_int32), ("obs_begin", C.c_int32 * (MPE_ROWS_MAX_ENTITIES + 1)),
        ("rew_begin", C.c_int32 * (MPE_ROWS_MAX_ENTITIES + 1)), ("n_vel", C.c_int32), ("n_regions", C.c_int32),
        ("region_entity", C.c_int32 * 2), ("all_seeing", C.c_uint32), ("image", C.c_void_p),
        ("done_begin", C.c_int32 * (MPE_ROWS_MAX_ENTITIES + 1)), ("reset_boxes", C.c_int32),
        ("reset_box", (C.c_float * 4) * MPE_ROWS_MAX_ENTITIES), ("n_shared", C.c_int32), ("traced", C.c_int32),
    ]


EXPORTS = {
    # name: (restype, argtypes)
    "mpe_abi_version": (C.c_int, []),
    "mpe_last_error": (C.c_char_p, []),
    "mpe_sizeof_desc": (C.c_size_t, []),
    "mpe_sizeof_buffers": (C.c_size_t, []),
    "mpe_fill_obs_layout": (C.c_int, [C.POINTER(MpeScenarioDesc)]),
    "mpe_fill_entity_table": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(C.c_float)]),
    "mpe_step_supported": (C.c_int, [C.POINTER(MpeScenarioDesc)]),
    "mpe_step": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_void_p]),
    "mpe_step_thread": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_void_p]),
    "mpe_observe": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_void_p]),
    "mpe_world_step": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_void_p]),
    "mpe_apply_action_force": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_void_p]),
    "mpe_collision_force": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_void_p]),
    "mpe_integrate_state": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_void_p]),
    "mpe_reset": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_void_p, C.c_float,
                            C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p]),
    "mpe_reset_random_actions_block": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_float, C.c_uint64,
                                                 C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, C.c_int64, C.c_void_p]),
    "mpe_reset_rows": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.POINTER(MpeRowProgram), C.c_int64, C.c_void_p,
                                 C.c_float, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p]),
    "mpe_random_actions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint64,
                                     C.c_int64, C.c_void_p]),
    "mpe_random_actions_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint64,
                                           C.c_int32, C.c_int64, C.c_void_p]),
    "mpe_random_comm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_uint32, C.c_uint64, C.c_uint64,
                                  C.c_int32, C.c_int64, C.c_void_p]),
    "mpe_rows_validate": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeRowProgram), C.POINTER(C.c_int32)]),
    "mpe_rows": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.POINTER(MpeRowProgram), C.c_int64, C.c_void_p]),
    "mpe_step_rows": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.POINTER(MpeRowProgram), C.c_int64, C.c_void_p]),
    "mpe_episode_finish": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.POINTER(MpeRowProgram), C.c_int64, C.c_void_p,
                                     C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p]),
    "mpe_step_rows_episode": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.POINTER(MpeRowProgram), C.c_int64, C.c_void_p,
                                        C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p]),
    "mpe_rollout_rows": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.POINTER(MpeRowProgram), C.c_int64, C.c_int32,
                                   C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_int64, C.c_int32, C.c_uint32, C.c_void_p]),
    "mpe_rollout_rows_episode": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.POINTER(MpeRowProgram), C.c_int64, C.c_int32,
                                           C.c_void_p, C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int32,
                                           C.c_uint32, C.c_void_p]),
    "mpe_sizeof_row_program": (C.c_size_t, []),
    "mpe_sizeof_step_server": (C.c_size_t, []),
    "mpe_rows_static_source": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeRowProgram), C.POINTER(C.c_int32), C.c_char_p,
                                         C.c_size_t, C.POINTER(C.c_size_t)]),
    "mpe_rows_load_image": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeRowProgram), C.POINTER(C.c_int32), C.c_void_p,
                                      C.c_size_t]),
    "mpe_rows_unload_image": (C.c_int, [C.POINTER(MpeRowProgram)]),
    "mpe_rows_image_active": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeRowProgram)]),
    "mpe_episode_tick": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "mpe_rollout_random": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_int32,
                                     C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_int64, C.c_int32,
                                     C.c_void_p]),
    "mpe_rollout_actions": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_int32, C.c_int32, C.c_float,
                                      C.c_uint64, C.c_uint64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "mpe_rollout_rows_actions": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.POINTER(MpeRowProgram), C.c_int64,
                                           C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_int64, C.c_int32, C.c_void_p,
                                           C.c_void_p]),
    "mpe_step_server_supported": (C.c_int, [C.POINTER(MpeScenarioDesc), C.c_int64]),
    "mpe_step_server_flags": (C.c_int64, [C.c_int64]),
    "mpe_step_server_start": (C.c_int, [C.POINTER(MpeScenarioDesc), C.POINTER(MpeBuffers), C.c_int64, C.c_int32, C.c_int32,
                                        C.c_float, C.c_uint64, C.c_uint64, C.c_int64, C.POINTER(MpeStepServer), C.c_void_p]),
    "mpe_step_server_ring": (C.c_int, [C.POINTER(MpeStepServer), C.c_uint64, C.c_void_p]),
    "mpe_step_server_wait": (C.c_int, [C.POINTER(MpeStepServer), C.c_int64, C.c_uint64, C.c_void_p]),
}

_lib = None


class MpeError(RuntimeError):
    pass


def lib():
    """Load libmpe_hip.so once.  torch is imported first so that the HIP runtime already mapped by
    torch (same SONAME libamdhip64.so.7) is the one our kernels register with."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MpeError(
            "%s is missing: build it with `python -m multiagent_particle_envs_amd._build` "
            "(there is no CPU fallback for the step path)" % LIB_PATH)
    import torch  # noqa: F401  (maps torch's libamdhip64 before ours resolves its NEEDED entry)
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(handle, name)  # AttributeError => symbol not exported
        fn.restype = res
        fn.argtypes = args
    if handle.mpe_abi_version() != MPE_ABI_VERSION:
        raise MpeError("ABI version mismatch: library %d, binding %d" % (handle.mpe_abi_version(), MPE_ABI_VERSION))
    if handle.mpe_sizeof_desc() != C.sizeof(MpeScenarioDesc) or handle.mpe_sizeof_buffers() != C.sizeof(MpeBuffers) or \
            handle.mpe_sizeof_row_program() != C.sizeof(MpeRowProgram) or handle.mpe_sizeof_step_server() != C.sizeof(MpeStepServer):
        raise MpeError("struct layout mismatch between include/mpe_hip.h and _abi.py")
    _lib = handle
    return _lib


def raw_stream(device):
    """The current HIP stream of `device` as a ctypes void* (what every entry point takes).  Uses torch's
    raw-handle accessor when it exists: ~0.3 us instead of ~2.5 us for torch.cuda.current_stream(), which
    matters when a step is a 6 us kernel."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    get = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if get is not None:
        return C.c_void_p(get(idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check(rc, what=""):
    if rc != 0:
        msg = lib().mpe_last_error().decode("utf-8", "replace")
        raise MpeError("%s failed (code %d): %s" % (what or "libmpe_hip call", rc, msg))
