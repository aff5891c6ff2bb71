"""ctypes binding of libmpe_hip.so (the C ABI declared in include/mpe_hip.h).

There is no CPU fallback: if the shared library is missing the import fails with
instructions to build it, and every non-zero return code of the library is raised
as :class:`MpeError` carrying ``mpe_last_error``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPE_HIP_LIB selects another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("MPE_HIP_LIB") or os.path.join(_HERE, "libmpe_hip.so")

MPE_F32, MPE_F64 = 0, 1
SCN_NONE, SCN_SPREAD, SCN_TAG, SCN_WORLD_COMM, SCN_SIMPLE, SCN_ADVERSARY, SCN_PUSH = 0, 1, 2, 3, 4, 5, 6
SCN_SPEAKER_LISTENER, SCN_REFERENCE, SCN_CRYPTO, SCN_PREDATOR_PREY, SCN_CONSENSUS = 7, 8, 9, 10, 11
SCN_COLLECTOR = 12
PHYSICS_CORE, PHYSICS_NEW_CORE = 0, 1
GAE_REFERENCE, GAE_STANDARD = 0, 1
MAX_AGENTS = 64
MAX_LANDMARKS = 64
MAX_WALLS = 16

_p_f64 = C.POINTER(C.c_double)
_p_u8 = C.POINTER(C.c_uint8)


class MpeSpec(C.Structure):
    _fields_ = [
        ("scenario", C.c_int32), ("n_agents", C.c_int32), ("n_landmarks", C.c_int32), ("dim_c", C.c_int32),
        ("dt", C.c_double), ("damping", C.c_double), ("contact_force", C.c_double), ("contact_margin", C.c_double),
        ("size", _p_f64), ("mass", _p_f64), ("movable", _p_u8), ("collide", _p_u8),
        ("sensitivity", _p_f64), ("max_speed", _p_f64), ("silent", _p_u8), ("adversary", _p_u8), ("leader", _p_u8),
        ("n_walls", C.c_int32), ("collaborative", C.c_int32),
        ("wall_vertical", _p_u8), ("wall_axis_pos", _p_f64), ("wall_end0", _p_f64), ("wall_end1", _p_f64),
        ("wall_width", _p_f64), ("wall_hard", _p_u8), ("ghost", _p_u8),
        ("physics", C.c_int32), ("reserved", C.c_int32), ("accel", _p_f64),
        ("u_noise", _p_f64), ("c_noise", _p_f64),
    ]


class MpeLayout(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int64), ("n_agents", C.c_int32), ("n_landmarks", C.c_int32), ("dim_c", C.c_int32),
        ("n_speakers", C.c_int32), ("n_act", C.c_int32), ("obs_total", C.c_int32),
        ("n_info_i", C.c_int32), ("n_info_f", C.c_int32), ("state_env_major", C.c_int32), ("n_aux", C.c_int32),
        ("obs_dim", C.c_int32 * MAX_AGENTS), ("obs_offset", C.c_int32 * MAX_AGENTS),
        ("n_hist_rows", C.c_int32), ("reserved", C.c_int32),
    ]


class MpeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mpe_hip error %d: %s" % (code, message))
        self.code = code


EXPORTS = {
    # name: (restype, argtypes)
    "mpe_create": (C.c_int, [C.POINTER(MpeSpec), C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mpe_destroy": (C.c_int, [C.c_void_p]),
    "mpe_layout": (C.c_int, [C.c_void_p, C.POINTER(MpeLayout)]),
    "mpe_last_error": (C.c_char_p, [C.c_void_p]),
    # (handle, actions, pos, vel, comm, aux, obs, rew, done, info_i, info_f, stream)
    "mpe_step": (C.c_int, [C.c_void_p] + [C.c_void_p] * 10 + [C.c_void_p]),
    # (..., info_f, step_count, episode_count, ep_return, last_return, max_steps, seed, stream)
    "mpe_step_auto": (C.c_int, [C.c_void_p] + [C.c_void_p] * 14 + [C.c_int32, C.c_uint64, C.c_void_p]),
    "mpe_rollout": (C.c_int, [C.c_void_p] + [C.c_void_p] * 10 + [C.c_int32, C.c_void_p]),
    # (handle, seed, episode, actions, pos, vel, comm, aux, obs, rew, done, info_i, info_f, obs0, n_steps, stream)
    "mpe_episode": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64] + [C.c_void_p] * 11 + [C.c_int32, C.c_void_p]),
    # (handle, seed, first_episode, n_episodes, steps_per_episode, actions, pos, vel, comm, aux, obs, rew, done, info_i, info_f, obs0, stream)
    "mpe_episodes": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32] + [C.c_void_p] * 11 + [C.c_void_p]),
    # (handle, pos, vel, comm, aux, obs, rew, info_i, info_f, stream)
    "mpe_observe": (C.c_int, [C.c_void_p] + [C.c_void_p] * 8 + [C.c_void_p]),
    "mpe_post_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    # (handle, enabled)
    "mpe_set_post_step": (C.c_int, [C.c_void_p, C.c_int]),
    # (handle, bound)
    "mpe_set_done_bound": (C.c_int, [C.c_void_p, C.c_double]),
    # (handle, env_base)
    "mpe_set_env_base": (C.c_int, [C.c_void_p, C.c_uint64]),
    # (handle, seed, draws)
    "mpe_set_noise": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    # (handle, set, seed*, step*)
    "mpe_noise_state": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    # (handle, seed, episode, mask, pos, vel, comm, aux, stream)
    "mpe_reset": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64] + [C.c_void_p] * 5 + [C.c_void_p]),
    "mpe_world_step": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_void_p]),
    # (spec, n_envs, dtype, device, u, c, pos, vel, comm, stream): no handle, no allocation
    "mpe_world_step_static": (C.c_int, [C.POINTER(MpeSpec), C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_void_p]),
    # (device, mode, n_envs, n_agents, n_steps, rewards, values, dones, last_values, last_dones, gamma, lam, adv, ret, stream)
    "mpe_gae": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [C.c_double, C.c_double] + [C.c_void_p] * 3),
}

_lib = None


def load():
    """Load libmpe_hip.so once and declare the prototypes of every exported entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: the HIP library has not been built. Run `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950) at the repository root; there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in EXPORTS.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(handle, code):
    if code != 0:
        msg = load().mpe_last_error(handle)
        raise MpeError(code, msg.decode() if msg else "?")
