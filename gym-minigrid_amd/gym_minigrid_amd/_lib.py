"""ctypes binding of libmgx.so (include/mgx.h).  There is no Python or CPU fallback: if the
HIP library has not been built this raises, loudly, at first use."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
# MGX_LIB: another build of the same library (tuning / instrumented variants, tools/ab.sh); never a different implementation
SO_PATH = os.environ.get("MGX_LIB") or os.path.join(CSRC, "libmgx.so")

MGX_OK = 0
ERR_NAMES = {-1: "INVALID_ARG", -2: "INVALID_STATE", -3: "INVALID_ACTION", -4: "OUT_OF_BOUNDS",
             -5: "UNSUPPORTED", -6: "HIP", -7: "NO_LEVELGEN"}
OBS_PARTIAL, OBS_FULL, OBS_PARTIAL_ONEHOT, OBS_FULL_ONEHOT, OBS_FULL_ONEHOT_NOCOLOR, OBS_PARTIAL_FLAT, OBS_FULL_FLAT = 0, 1, 2, 3, 4, 5, 6
# mgx_task_kind (include/mgx.h)
(TASK_NONE, TASK_FETCH, TASK_GOTODOOR, TASK_DYNOBS, TASK_GOTOOBJECT, TASK_REDBLUEDOORS, TASK_MEMORY, TASK_UNLOCK, TASK_PICKUPBOX, TASK_NOTE,
 TASK_PUTNEAR, TASK_TWOGOALS) = range(12)
TASKS_WITH_EPISODE_MISSION = (TASK_FETCH, TASK_GOTOOBJECT, TASK_PICKUPBOX, TASK_NOTE, TASK_PUTNEAR)  # the mission names per-episode objects


class MgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mgx error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


class InvalidAction(MgxError, AssertionError):
    """The reference raises AssertionError('unknown action') (minigrid.py:1316-1318)."""


class OutOfBounds(MgxError, AssertionError):
    """The reference's Grid.get asserts its bounds (minigrid.py:416-419)."""


class Config(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("max_steps", ctypes.c_int32),
                ("see_through_walls", ctypes.c_int32), ("lava_v1", ctypes.c_int32), ("obs_mode", ctypes.c_int32),
                ("auto_reset", ctypes.c_int32), ("level_kind", ctypes.c_int32), ("level_arg0", ctypes.c_int32),
                ("level_arg1", ctypes.c_int32), ("new_level_each_episode", ctypes.c_int32), ("agent_view_size", ctypes.c_int32),
                ("extended_actions", ctypes.c_int32), ("alt_visibility", ctypes.c_int32),
                ("task_kind", ctypes.c_int32), ("object_state", ctypes.c_int32)]


class Stats(ctypes.Structure):
    _fields_ = [("steps", ctypes.c_int64), ("episodes", ctypes.c_int64), ("reward_sum", ctypes.c_double),
                ("invalid_actions", ctypes.c_int64), ("out_of_bounds", ctypes.c_int64)]


_vp, _i64, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
SIGNATURES = {
    "mgx_last_error": (ctypes.c_char_p, []),
    "mgx_version": (ctypes.c_char_p, []),
    "mgx_env_config": (_int, [ctypes.c_char_p, ctypes.POINTER(Config)]),
    "mgx_env_id": (ctypes.c_char_p, [_int]),
    "mgx_mission": (_int, [ctypes.POINTER(Config), ctypes.c_uint32, ctypes.c_char_p, _int]),
    "mgx_create": (_int, [ctypes.POINTER(Config), _i64, _int, ctypes.POINTER(_vp)]),
    "mgx_destroy": (_int, [_vp]),
    "mgx_set_stream": (_int, [_vp, _vp]),
    "mgx_use_own_stream": (_int, [_vp]),
    "mgx_sync": (_int, [_vp]),
    "mgx_clear_faults": (_int, [_vp]),
    "mgx_obs_bytes": (_int, [_vp, ctypes.POINTER(_i64)]),
    "mgx_generate_levels": (_int, [ctypes.POINTER(Config), _i64, _vp, _vp, _vp]),
    "mgx_generate_level_stream_ex": (_int, [ctypes.POINTER(Config), ctypes.c_uint64, _i64, _vp, _vp, _vp]),
    "mgx_generate_levels_ex": (_int, [ctypes.POINTER(Config), _i64, _vp, _vp, _vp, _vp]),
    "mgx_generate_levels_full": (_int, [ctypes.POINTER(Config), _i64, _vp, _vp, _vp, _vp, _vp]),
    "mgx_generate_level_stream_full": (_int, [ctypes.POINTER(Config), ctypes.c_uint64, _i64, _vp, _vp, _vp, _vp]),
    "mgx_rollout": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "mgx_set_task": (_int, [_vp, _vp]),
    "mgx_get_task": (_int, [_vp, _vp]),
    "mgx_generate_level_stream": (_int, [ctypes.POINTER(Config), ctypes.c_uint64, _i64, _vp, _vp]),
    "mgx_reset": (_int, [_vp, _vp, _vp, _vp]),
    "mgx_set_seed_schedule": (_int, [_vp, _vp, ctypes.c_int32, ctypes.c_int32]),
    "mgx_add_bonus": (_int, [_vp, ctypes.c_int32]),
    "mgx_set_dac": (_int, [_vp, ctypes.c_int32]),
    "mgx_get_bonus_counts": (_int, [_vp, ctypes.c_int32, _vp]),
    "mgx_set_state": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mgx_get_state": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mgx_set_object_state": (_int, [_vp, _vp, _vp, _vp]),
    "mgx_get_object_state": (_int, [_vp, _vp, _vp, _vp]),
    "mgx_observe": (_int, [_vp, _vp]),
    "mgx_get_direction": (_int, [_vp, _vp]),
    "mgx_get_pose": (_int, [_vp, _vp]),
    "mgx_step": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "mgx_get_stats": (_int, [_vp, ctypes.POINTER(Stats)]),
    "mgx_read_stats_async": (_int, [_vp, _vp]),
    "mgx_fill_actions": (_int, [_vp, ctypes.c_uint64, _i64, _i64, _i64, _vp]),
    "mgx_action_at": (ctypes.c_uint32, [ctypes.c_uint64, _i64, _i64]),
    "mgx_step_kernel_name": (_int, [_vp, ctypes.c_char_p, _int]),
    "mgx_profile_begin": (_int, [_vp]),
    "mgx_profile_begin_sampled": (_int, [_vp, _int]),
    "mgx_profile_stop": (_int, [_vp]),
    "mgx_profile_end": (_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double)]),
    "mgx_profile_kernel": (_int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


def lib():
    """Load libmgx.so (built by `make -C gym-minigrid_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("libmgx.so is missing (%s). Build it with `make -C %s` or "
                              "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback."
                              % (SO_PATH, CSRC))
        # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64.so.7 / libhsa-runtime64.
        # Importing torch first makes libmgx.so's NEEDED libamdhip64.so.7 resolve (by SONAME) to that already
        # loaded copy, so torch tensors and libmgx share one runtime, one context and one address space.  Loading
        # in the other order puts two runtimes in the process and torch then finds "No HIP GPUs".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != MGX_OK:
        msg = lib().mgx_last_error().decode()
        if rc == -3:
            raise InvalidAction(rc, msg)
        if rc == -4:
            raise OutOfBounds(rc, msg)
        raise MgxError(rc, msg)


def env_config(env_id):
    c = Config()
    check(lib().mgx_env_config(env_id.encode(), ctypes.byref(c)))
    return c


def env_ids():
    out, i = [], 0
    while True:
        s = lib().mgx_env_id(i)
        if s is None:
            return out
        out.append(s.decode())
        i += 1
