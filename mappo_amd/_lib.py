"""ctypes binding of libmappo_hip.so (C ABI: include/mappo_hip.h).  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os

import torch  # noqa: F401  — must come first: libmappo_hip.so has to bind to the HIP runtime torch already loaded
#               (loading /opt/rocm's libamdhip64 before torch's own copy gives a process with two runtimes, and
#               launches from the second one fail with "no ROCm-capable device is detected")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAPPO_HIP_LIB") or os.path.join(_HERE, "libmappo_hip.so")     # override: diagnostic builds only

HIDDEN = 64
MAX_ACTIONS = 32
MAX_LAYER_N = 2


class NetDesc(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("hidden", C.c_int32), ("out_dim", C.c_int32), ("layer_N", C.c_int32),
                ("use_relu", C.c_int32), ("use_feature_norm", C.c_int32), ("recurrent", C.c_int32)]


class PpoCfg(C.Structure):
    _fields_ = [("clip_param", C.c_float), ("entropy_coef", C.c_float), ("value_loss_coef", C.c_float),
                ("huber_delta", C.c_float), ("use_huber_loss", C.c_int32), ("use_clipped_value_loss", C.c_int32),
                ("use_policy_active_masks", C.c_int32), ("use_value_active_masks", C.c_int32),
                ("use_valuenorm", C.c_int32), ("accumulate_partials", C.c_int32)]


class SmacSlot(C.Structure):
    """mappo_smac_slot: destination arrays of one buffer slot for the fused SMAC insert (include/mappo_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("obs", "share_obs", "available_actions", "rewards", "masks", "bad_masks", "active_masks",
                                          "rnn_states", "rnn_states_critic")]


_P, _I32, _I64, _F, _D, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_uint64

# name -> (restype, argtypes); must list every symbol include/mappo_hip.h declares (tests/test_capi_symbols.py)
SIGNATURES = {
    "mappo_last_error": (C.c_char_p, []),
    "mappo_abi_version": (C.c_int, []),
    "mappo_net_param_count": (_I64, [C.POINTER(NetDesc)]),
    "mappo_insert_mpe": (C.c_int, [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "mappo_insert_mpe_rnn": (C.c_int, [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P,
                                       _I32, _P]),
    "mappo_insert_smac": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _I64, _I64, _P, _P, _P] + [_P] * 9 + [_I32] * 6 + [_P]),
    "mappo_recurrent_rows": (C.c_int, [_P, _I32, _I64, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "mappo_copy_batch": (C.c_int, [_I32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(_I64), _P]),
    "mappo_gae_scan": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _F, _F, _I32, _I32, _P]),
    "mappo_adv_workspace_bytes": (_I64, [_I64]),
    "mappo_adv_moments": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mappo_adv_normalize": (C.c_int, [_P, _P, _I64, _P]),
    "mappo_moments_workspace_bytes": (_I64, [_I64]),
    "mappo_minibatch_moments": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P]),
    "mappo_valuenorm_update": (C.c_int, [_P, _P, _D, _P]),
    "mappo_valuenorm_update_n": (C.c_int, [_P, _P, _D, _I32, _P, _P]),
    "mappo_ppo_loss_workspace_bytes": (_I64, [_I64]),
    "mappo_ppo_loss_fwd_bwd": (C.c_int, [_P] * 16 + [C.POINTER(PpoCfg), _I64, _I32, _P]),
    "mappo_mlp_forward": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I64, _P, _P]),
    "mappo_actor_act": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I64, _I32, _U64, _U64, _P, _P, _P, _P]),
    "mappo_rollout_step": (C.c_int, [_P, C.POINTER(NetDesc), _P, C.POINTER(NetDesc), _P, _I64, _I64, _P, _I64, _I64, _I32, _I64, _P, _I32,
                                     _U64, _U64, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _I64, _I64, _P, _P, _I32, _P]),
    "mappo_mlp_backward_slabs": (_I32, [_I64]),
    "mappo_mlp_backward": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I64, _P, _P, _I64, _I64, _P, _P]),
    "mappo_wide_workspace_floats": (_I64, [_I64]),
    "mappo_wide_l1_slabs": (_I32, [_I64]),
    "mappo_wide_layout": (_I32, [C.POINTER(NetDesc), _I32]),
    "mappo_wide_l1_backward": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I64, _P, _P, _I64, _I64, _I32, _P]),
    "mappo_update_partials_bytes": (_I64, []),
    "mappo_actor_update": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I64, _P, _P, _P, _P, _P, _P, C.POINTER(PpoCfg), _P, _I64,
                                     _I64, _P, _P, _I32, _P]),
    "mappo_critic_update": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I64, _P, _P, _P, _P, _P, C.POINTER(PpoCfg), _P, _I64,
                                      _I64, _P, _P, _I32, _P]),
    "mappo_dual_update_slabs": (_I32, [C.POINTER(NetDesc), C.POINTER(NetDesc), _I64]),
    "mappo_actor_critic_update": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, C.POINTER(NetDesc), _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                            C.POINTER(PpoCfg), _P, _I64, _I64, _I64, _P, _P, _P]),
    "mappo_update_stats": (C.c_int, [_P, _I32, _P, _I32, _P, C.POINTER(PpoCfg), _P, _P, _P]),
    "mappo_mlp_features": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I64, _P, _P]),
    "mappo_gru_forward": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _P, _P, _P, _I32, _I32, _P, _I32, _P, _P, _I32, _U64, _U64,
                                    _P, _P, _P, _P]),
    "mappo_gru_step_dual": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _P, _P, C.POINTER(NetDesc), _P, _P, _P, _P, _I32, _P, _I32, _U64, _U64,
                                      _P, _P, _P, _P, _P]),
    "mappo_recurrent_step_dual": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _P, _P, C.POINTER(NetDesc), _P, _P, _P, _P, _I32, _P, _I32, _U64, _U64,
                                      _P, _P, _P, _P, _P]),
    "mappo_gru16_scratch_floats": (_I64, [_I32, _I32]),
    "mappo_gru16_blocked_floats": (_I64, [_I32, _I32]),
    "mappo_gru16_slabs": (_I32, [_I32, _I32]),
    "mappo_mlp_features_seq": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I32, _I32, _P, _P]),
    "mappo_gru16_forward_loss": (C.c_int, [_P, C.POINTER(NetDesc), _P, _I32, _P, _P, _P, _P, _I32, _I32, _I32] + [_P] * 9 +
                                 [C.POINTER(PpoCfg), _P, _P, _I64, _I64, _P, _P]),
    "mappo_gru16_backward": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I32, _I32, _P, _P, _P]),
    "mappo_gru16_wgrad": (C.c_int, [C.POINTER(NetDesc), _P, _I32, _P, _I32, _I32, _P, _I64, _I64, _P]),
    "mappo_trunk_backward_seq": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I32, _I32, _P, _P, _I64, _I64, _P, _P]),
    "mappo_recurrent_rollout_step": (C.c_int, [_P, C.POINTER(NetDesc), _P, C.POINTER(NetDesc), _P, _P, _P, _P, _I64, _I64, _P, _I64, _I64, _P,
                                               _P, _P, _P, _P, _I32, _I32, _I32, _U64, _U64, _P, _P, _P, _P, C.POINTER(SmacSlot), _P]),
    "mappo_mlp_features_dual": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _P, C.POINTER(NetDesc), _P, _P, _I64, _P]),
    "mappo_trunk_backward": (C.c_int, [_P, C.POINTER(NetDesc), _P, _P, _I64, _P, _P, _I64, _I64, _P, _P]),
    "mappo_optim_workspace_bytes": (_I64, [_I64]),
    "mappo_slab_reduce": (C.c_int, [_P, _I32, _I64, _I64, _P, _P]),
    "mappo_clip_adam": (C.c_int, [_P, _P, _P, _P, C.POINTER(_I64), _I32, _P, _P, _P, _P, _P, _P]),
    "mappo_reduce_clip_adam": (C.c_int, [_P, _I32, _I64, _P, _P, _P, _P, C.POINTER(_I64), _I32, _P, _P, _P, _P, _P, _P]),
    "mappo_mpe_spread_reset": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _U64, _P]),
    "mappo_synth_smac_pool": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, C.c_float, C.c_float, _U64, _P, _P]),
    "mappo_synth_smac_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, C.c_float, C.c_float, _U64, _P, _P]),
    "mappo_mpe_spread_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _I32, _I32, _I32, _I32, _U64, _P]),
    "mappo_profile_arm": (C.c_int, [_I32, _P, _P]),
    "mappo_selftest_mfma": (C.c_int, [_P, _P, _P, _P]),
}

_lib = None


class MappoHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built: `python -m mappo_amd.build`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MappoHipError(f"{LIB_PATH} is missing — build it with `python -m mappo_amd.build` "
                            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the MAPPO hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mappo_last_error().decode("utf-8", "replace")
        raise MappoHipError(f"{what} failed (code {rc}): {msg}")
