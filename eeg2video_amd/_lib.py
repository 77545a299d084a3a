"""ctypes binding of ``libeeg2video_hip.so`` (include/eeg2video_hip.h, include/eeg2video_hip_ops.h).

There is no fallback: if the shared library has not been built (``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C eeg2video_amd/csrc``) importing the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("E2V_LIB_PATH") or os.path.join(_HERE, "lib", "libeeg2video_hip.so")   # override: A/B of two builds

c_int64_p = C.POINTER(C.c_int64)
c_float_p = C.c_void_p          # device pointers travel as integers (tensor.data_ptr())


class E2VConfig(C.Structure):
    """``e2v_config`` of include/eeg2video_hip.h (field order is the ABI)."""
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int),
        ("block_out_channels", C.c_int * 4),
        ("layers_per_block", C.c_int), ("cross_attention_dim", C.c_int), ("attention_heads", C.c_int),
        ("norm_num_groups", C.c_int), ("norm_eps", C.c_float), ("flip_sin_to_cos", C.c_int), ("freq_shift", C.c_float),
        ("vae_in_channels", C.c_int), ("vae_latent_channels", C.c_int),
        ("vae_block_out_channels", C.c_int * 4),
        ("vae_layers_per_block", C.c_int), ("vae_norm_num_groups", C.c_int), ("vae_norm_eps", C.c_float),
        ("vae_scaling_factor", C.c_double),
        ("num_train_timesteps", C.c_int), ("beta_start", C.c_double), ("beta_end", C.c_double),
        ("steps_offset", C.c_int),
        ("sem_in_features", C.c_int), ("sem_hidden", C.c_int), ("sem_tokens", C.c_int),
        ("attention_heads_per_block", C.c_int * 4),
    ]


E2V_OK, E2V_EINVAL, E2V_ESHAPE, E2V_ENOWEIGHT, E2V_EHIP, E2V_ESTATE = 0, -1, -2, -3, -4, -5
E2V_F32, E2V_F16, E2V_BF16, E2V_F32X3 = 0, 1, 2, 3

_ctx = C.c_void_p
_stream = C.c_void_p
_i, _f, _i64, _p = C.c_int, C.c_float, C.c_int64, C.c_void_p

#: every symbol the two headers declare: name -> (restype, argtypes)
SIGNATURES = {
    "e2v_default_config": (None, [C.POINTER(E2VConfig)]),
    "e2v_config_size": (_i64, []),
    "e2v_version": (C.c_char_p, []),
    "e2v_create": (_i, [C.POINTER(E2VConfig), _i, C.POINTER(_ctx)]),
    "e2v_destroy": (None, [_ctx]),
    "e2v_last_error": (C.c_char_p, [_ctx]),
    "e2v_load_tensor": (_i, [_ctx, C.c_char_p, _p, _i, c_int64_p, _i]),
    "e2v_num_expected_keys": (_i64, [_ctx]),
    "e2v_expected_key": (C.c_char_p, [_ctx, _i64, c_int64_p, C.POINTER(_i)]),
    "e2v_finalize_weights": (_i, [_ctx, _i]),
    "e2v_ddim_timesteps": (_i, [_ctx, _i, c_int64_p]),
    "e2v_ddim_alphas_cumprod": (_i, [_ctx, C.POINTER(C.c_float)]),
    "e2v_set_alphas_cumprod": (_i, [_ctx, C.POINTER(C.c_float), _i]),
    "e2v_set_ddim_schedule": (_i, [_ctx, C.POINTER(C.c_float), _i, _i]),
    "e2v_unet_forward": (_i, [_ctx, _p, c_int64_p, _i, _p, _i, _i, _i, _i, _i, _p, _stream]),
    "e2v_unet_forward_ft": (_i, [_ctx, _p, C.POINTER(C.c_float), _i, _p, _i, _i, _i, _i, _i, _p, _stream]),
    "e2v_ddim_cfg_step": (_i, [_ctx, _p, _p, _p, _p, _i64, _f, _i64, _i64, _stream]),
    "e2v_vae_decode": (_i, [_ctx, _p, _i, _i, _i, _i, _i, _p, _stream]),
    "e2v_vae_encode": (_i, [_ctx, _p, _i, _i, _i, _p, _p, _stream]),
    "e2v_generate": (_i, [_ctx, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p, _p, _stream]),
    "e2v_semantic_predict": (_i, [_ctx, _p, _i, _p, _stream]),
    "e2v_dana_noise": (_i, [_ctx, _p, _p, _p, c_int64_p, _i, _f, _i, _i, _i, _i, _i, _p, _stream]),
    "e2v_frames_to_uint8": (_i, [_ctx, _p, _p, _i64, _stream]),
    "e2v_cfg_combine": (_i, [_ctx, _p, _p, _f, _p, _i64, _stream]),
    "e2v_lincomb": (_i, [_ctx, _i, C.POINTER(C.c_void_p), C.POINTER(C.c_float), _p, _i64, _stream]),
    "e2v_ddim_next_step": (_i, [_ctx, _p, _p, _p, _i64, _i64, _i, _stream]),
    "e2v_ddim_invert": (_i, [_ctx, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _stream]),
    "e2v_set_compute_dtype": (_i, [_ctx, _i]),
    "e2v_set_conv_algo": (_i, [_ctx, _i]),
    "e2v_device_bytes": (_i64, [_ctx]),
    "e2v_comm_unique_id": (_i, [_p]),
    "e2v_comm_init": (_i, [_ctx, _p, _i, _i]),
    "e2v_comm_world": (_i, [_ctx]),
    "e2v_allgather_frames": (_i, [_ctx, _p, _i64, _i, _p, _stream]),
    "e2v_comm_destroy": (_i, [_ctx]),
    "e2v_profile_begin": (_i, [_ctx]),
    "e2v_profile_end": (_i64, [_ctx, C.c_char_p, _i64]),
    # eeg2video_hip_ops.h
    "e2v_op_conv3x3": (_i, [_ctx, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _p, _stream]),
    "e2v_op_linear": (_i, [_ctx, _p, _i, _i64, _i, _p, _p, _i, _p, _i, _p, _stream]),
    "e2v_op_groupnorm": (_i, [_ctx, _p, _i, _p, _i, _i, _i, _i, _f, _p, _p, _i, _p, _stream]),
    "e2v_op_layernorm": (_i, [_ctx, _p, _i64, _i, _p, _p, _f, _p, _stream]),
    "e2v_op_attention": (_i, [_ctx, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _stream]),
    "e2v_op_temporal_attention": (_i, [_ctx, _p, _p, _i, _i, _i, _i, _i, _f, _stream]),
    "e2v_op_to_channels_last": (_i, [_ctx, _p, _p, _i, _i, _i, _i, _stream]),
    "e2v_op_from_channels_last": (_i, [_ctx, _p, _i, _p, _i, _i, _i, _stream]),
    "e2v_op_set_knob": (_i, [C.c_char_p, _i]),
    "e2v_op_rowblock_sums": (_i, [_ctx, _p, _i64, _i, _p, _stream]),
    "e2v_op_describe_dispatch": (_i, [_ctx, _i, _i, _i, _i, _i, _i, C.c_char_p, _i64, c_int64_p]),
    "e2v_op_unet_forward_taps": (_i, [_ctx, _p, c_int64_p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i64, c_int64_p, C.POINTER(_i), _stream]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises ``ImportError`` if it is missing -- no CPU path exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built. "
            "Run `make -C eeg2video_amd/csrc` (or __graft_entry__.build()). There is no CPU fallback.")
    # torch ships its own libamdhip64 with the same SONAME; importing torch first makes the loader reuse it,
    # so that torch tensors and this library share one HIP runtime.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is only plumbing
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.e2v_config_size() != C.sizeof(E2VConfig):
        raise ImportError(f"{LIB_PATH}: e2v_config is {lib.e2v_config_size()} bytes in the library, {C.sizeof(E2VConfig)} in this binding "
                          "(include/eeg2video_hip.h and eeg2video_amd/_lib.py disagree)")
    _lib = lib
    return lib
