"""ctypes binding of include/rsrl_hip.h.  Fails loudly when the HIP library is missing: there is no
CPU fallback anywhere in this package."""
import ctypes as C
import os

from . import _build

f32p, i32p, u8p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("domain", C.c_int32), ("basis", C.c_int32),
        ("order", C.c_int32), ("n_tilings", C.c_int32), ("tiles_per_dim", C.c_int32), ("algo", C.c_int32),
        ("policy", C.c_int32), ("weight_mode", C.c_int32), ("weight_dtype", C.c_int32),
        ("max_episode_steps", C.c_uint32), ("n_envs", C.c_int64), ("env_offset", C.c_int64),
        ("seed", C.c_uint64), ("gamma", C.c_double), ("lr", C.c_double), ("alpha", C.c_double),
        ("epsilon", C.c_double), ("tau", C.c_double), ("steps_per_launch", C.c_uint32),
        ("trace", C.c_int32), ("stream", C.c_void_p), ("lam", C.c_double),
        ("lr_td", C.c_double),
        ("agent_policy", C.c_int32), ("exchange", C.c_int32), ("agent_epsilon", C.c_double), ("agent_tau", C.c_double),
        ("sigma", C.c_double), ("n_steps", C.c_int32), ("peer_timeout_ms", C.c_int32),
        ("epsilon_decay", C.c_double), ("epsilon_min", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [("env_steps", C.c_uint64), ("episodes", C.c_uint64), ("episodes_truncated", C.c_uint64),
                ("sum_episode_steps", C.c_uint64), ("sum_abs_td_error", C.c_double), ("sum_reward", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/rsrl_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "rsrl_hip_abi_version": (C.c_int, []),
    "rsrl_hip_device_count": (C.c_int, []),
    "rsrl_hip_last_error": (C.c_char_p, []),
    "rsrl_hip_config_init": (C.c_int, [C.POINTER(Config)]),
    "rsrl_hip_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "rsrl_hip_destroy": (C.c_int, [C.c_void_p]),
    "rsrl_hip_sync": (C.c_int, [C.c_void_p]),
    "rsrl_hip_state_dim": (C.c_int, [C.c_void_p]),
    "rsrl_hip_n_actions": (C.c_int, [C.c_void_p]),
    "rsrl_hip_n_outputs": (C.c_int, [C.c_void_p]),
    "rsrl_hip_n_features": (C.c_int, [C.c_void_p]),
    "rsrl_hip_n_envs": (C.c_int64, [C.c_void_p]),
    "rsrl_hip_state_bounds": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rsrl_hip_reset": (C.c_int, [C.c_void_p]),
    "rsrl_hip_get_states": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrl_hip_set_states": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrl_hip_get_actions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrl_hip_set_actions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrl_hip_get_episode_steps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrl_hip_set_episode_steps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrl_hip_get_q_carry": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "rsrl_hip_set_q_carry": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrl_hip_domain_step": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5),
    "rsrl_hip_domain_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrl_hip_q_evaluate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_q_find_max": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rsrl_hip_q_find_min": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rsrl_hip_q_expected_value": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rsrl_hip_policy_prob": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_project": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_tile_indices": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_handle": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]),
    "rsrl_hip_policy_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_policy_mode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_policy_probs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_set_epsilon": (C.c_int, [C.c_void_p, C.c_double]),
    "rsrl_hip_get_epsilons": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrl_hip_get_weights": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_set_weights": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_set_weights_all": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsrl_hip_save_weights": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rsrl_hip_load_weights": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rsrl_hip_get_traces": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_set_traces": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_get_td_weights": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_set_td_weights": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "rsrl_hip_train": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(Stats)]),
    "rsrl_hip_step_count": (C.c_uint64, [C.c_void_p]),
    "rsrl_hip_pending_steps": (C.c_int64, [C.c_void_p]),
    "rsrl_hip_rollout_greedy": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rsrl_hip_rollout_trajectory": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 6),
    "rsrl_hip_rollout_policy": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int64, C.c_int64] + [C.c_void_p] * 6),
    "rsrl_hip_checksum": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rsrl_hip_fx_saturations": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rsrl_hip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "rsrl_hip_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rsrl_hip_peer_export": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "rsrl_hip_can_access_peer": (C.c_int, [C.c_int, C.c_int]),
    "rsrl_hip_device_identity": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "rsrl_hip_peer_connect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rsrl_hip_group_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "rsrl_hip_group_train": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64]),
    "rsrl_hip_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rsrl_hip_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "rsrl_hip_timing_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_char_p)]),
    "rsrl_hip_measure_copy": (C.c_int, [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
}

_lib = None


class RsrlHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rsrl_hip error {code}: {msg}")
        self.code = code


def lib():
    """Load rsrl_amd/lib/librsrl_hip.so (built by __graft_entry__.build()); no fallback."""
    global _lib
    if _lib is None:
        path = os.environ.get("RSRL_HIP_LIB", _build.LIB_PATH)   # override: A/B kernel variants only
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). rsrl_amd has no CPU fallback.")
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the ABI header and the library drift apart
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RsrlHipError(rc, (lib().rsrl_hip_last_error() or b"").decode())
