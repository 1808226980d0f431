"""ctypes binding of libhiddenpose_hip.so (C ABI: include/hiddenpose_hip.h).

The library is REQUIRED: there is no eager/CPU fallback anywhere in this package.
`lib()` raises if the shared object has not been built (python -m hiddenpose_amd.build).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libhiddenpose_hip.so")

_lock = threading.Lock()
_lib = None


class HiddenPoseHipError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol declared in include/hiddenpose_hip.h
_vp, _fp, _i, _sz, _d = C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_double
SIGNATURES = {
    "hp_version": (_i, []),
    "hp_last_error_string": (C.c_char_p, []),
    "hp_profile_enable": (_i, [_i]),
    "hp_profile_reset": (_i, []),
    "hp_profile_count": (_i, []),
    "hp_profile_get": (_i, [_i, C.c_char_p, _i, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "hp_range_enable": (_i, [_i]),
    "hp_range_push": (_i, [C.c_char_p]),
    "hp_range_pop": (_i, []),
    "hp_range_start": (C.c_int64, [C.c_char_p]),
    "hp_range_stop": (_i, [C.c_int64]),
    "hp_lct_host_constants": (_i, [_i, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hp_lct_plan_create": (_i, [C.POINTER(_vp), _i, _i, _d, _d, _i, _i]),
    "hp_lct_plan_create_mode": (_i, [C.POINTER(_vp), _i, _i, _d, _d, _i, _i, _i]),
    "hp_lct_plan_destroy": (_i, [_vp]),
    "hp_laplacian5_forward": (_i, [_fp, _fp, _fp, C.c_long, _i, _i, _i, _vp]),
    "hp_laplacian5_backward": (_i, [_fp, _fp, _fp, C.c_long, _i, _i, _i, _vp]),
    "hp_lct_workspace_bytes": (_sz, [_vp, _i]),
    "hp_lct_forward": (_i, [_vp, _fp, _fp, _i, _vp, _sz, _vp]),
    "hp_lct_backward": (_i, [_vp, _fp, _fp, _i, _vp, _sz, _vp]),
    "hp_lct_plan_get_invpsf": (_i, [_vp, _vp, _vp]),
    "hp_conv3d_packed_weight_elems": (_sz, [_vp]),
    "hp_conv3d_pack_weight": (_i, [_vp, _fp, _fp, _fp, _vp]),
    "hp_conv3d_unpack_wgrad": (_i, [_vp, _fp, _fp, _vp]),
    "hp_conv3d_forward": (_i, [_vp, _fp, _fp, _fp, _fp, _vp, _vp]),
    "hp_conv3d_backward_data": (_i, [_vp, _fp, _fp, _fp, _fp, _vp]),
    "hp_conv3d_backward_data_masked": (_i, [_vp, _fp, _fp, _fp, _fp, _vp, _vp]),
    "hp_conv3d_backward_data_bnsums": (_i, [_vp, _fp, _fp, _fp, _fp, _vp, _fp, _fp, _fp, _fp, _fp, _i, _vp, _vp, C.POINTER(C.c_int), _vp]),
    "hp_bn_backward_presummed": (_i, [_fp, _fp, _fp, C.c_long, _i, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _vp, _vp, _vp, _i, _vp]),
    "hp_conv3d_backward_weight": (_i, [_vp, _fp, _fp, _fp, _vp]),
    "hp_conv3d_backward_weight_split": (_i, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "hp_bn_train_finalize": (_i, [_vp, C.c_long, _i, C.c_float, C.c_float, _fp, _fp, _fp, _fp, _vp]),
    "hp_bn_train_finalize_counted": (_i, [_vp, C.c_long, _i, C.c_float, C.c_float, _fp, _fp, _fp, _fp, _vp, _vp]),
    "hp_bn_eval_stats": (_i, [_fp, _fp, _i, C.c_float, _fp, _fp, _vp]),
    "hp_bn_apply": (_i, [_fp, _fp, _fp, C.c_long, _i, _fp, _fp, _fp, _fp, _i, _vp, _i, _vp]),
    "hp_bn_apply_res_bn": (_i, [_fp, _fp, _fp, C.c_long, _i, _fp, _fp, _fp, _fp, _i, _vp, _fp, _fp, _fp, _fp, _i, _vp]),
    "hp_cast_f32_to_bf16": (_i, [_fp, _vp, C.c_long, _vp]),
    "hp_cast_bf16_to_f32": (_i, [_vp, _fp, C.c_long, _vp]),
    "hp_bn_backward_workspace_bytes": (_sz, [_i]),
    "hp_bn_backward_dual": (_i, [_fp, _vp, C.c_long, _i, _fp, _fp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i,
                                 _fp, _fp, _vp, _i, _vp]),
    "hp_bn_backward": (_i, [_fp, _fp, _fp, _fp, _fp, C.c_long, _i, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _vp, _vp, _i, _vp]),
    "hp_maxpool3d_k3s2_forward": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "hp_maxpool3d_k3s2_backward": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "hp_layout_transpose": (_i, [_fp, _fp, _i, C.c_long, _i, _i, _vp]),
    "hp_stem_bn_pool_workspace_bytes": (_sz, [_i]),
    "hp_stem_bn_relu_pool_forward": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _vp, _vp]),
    "hp_stem_bn_relu_pool_backward": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _i, _fp, _fp, _vp, _vp]),
    "hp_dconv3_forward": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "hp_dconv3_forward_fused": (_i, [_fp, _fp, _fp, _fp, _fp, _vp, _i, _i, _i, _i, _i, _i, _i, C.c_float, _vp]),
    "hp_dconv3_backward_data_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "hp_dconv3_backward_data": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "hp_dconv3_forward_fused_p": (_i, [_fp, _fp, _fp, _fp, _fp, _vp, _i, _i, _i, _i, _i, _i, _i, C.c_float, _i, _vp]),
    "hp_dconv3_backward_data_p": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "hp_dconv3_backward_weight_p": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "hp_dconv3_backward_weight_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "hp_dconv3_backward_weight": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "hp_groupnorm_workspace_bytes": (_sz, [_i, _i]),
    "hp_groupnorm_relu_forward": (_i, [_fp, _fp, _i, _i, _i, C.c_long, _fp, _fp, C.c_float, _fp, _fp, _vp, _vp]),
    "hp_groupnorm_relu_forward_v2": (_i, [_fp, _fp, _i, _i, _i, C.c_long, _fp, _fp, C.c_float, _vp, _fp, _fp, _fp, _fp, _vp, _vp]),
    "hp_groupnorm_relu_backward_v2": (_i, [_fp, _fp, _fp, _i, _i, _i, C.c_long, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _vp]),
    "hp_maxpool3d_k2_forward": (_i, [_fp, _fp, C.c_long, _i, _i, _i, _vp]),
    "hp_maxpool3d_k2_backward": (_i, [_fp, _fp, _fp, C.c_long, _i, _i, _i, _vp]),
    "hp_maxpool3d_k2_backward_add": (_i, [_fp, _fp, _fp, C.c_long, _fp, _i, _i, _i, _i, _i, _vp]),
    "hp_upsample_trilinear2x_forward": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "hp_upsample_trilinear2x_forward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "hp_upsample_trilinear2x_forward_ws": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "hp_upsample_trilinear2x_backward": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "hp_upsample_trilinear2x_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "hp_upsample_trilinear2x_backward_ws": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "hp_channel_slice_copy": (_i, [_fp, _fp, _i, _i, C.c_long, _i, _i, _i, _vp]),
    "hp_conv1x1_forward": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, C.c_long, _vp]),
    "hp_conv1x1_backward": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, C.c_long, _vp]),
    "hp_conv1x1_forward_sum": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, C.c_long, _vp]),
    "hp_conv1x1_backward_sum": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, C.c_long, _vp]),
    "hp_leaky_add_forward": (_i, [_fp, _fp, _fp, C.c_long, C.c_float, _vp]),
    "hp_leaky_backward": (_i, [_fp, _fp, _fp, C.c_long, C.c_float, _vp]),
    "hp_normalize_feature_forward": (_i, [_fp, _fp, _i, C.c_long, C.c_float, _vp, _vp]),
    "hp_normalize_feature_backward": (_i, [_fp, _fp, _fp, _i, C.c_long, C.c_float, _vp, _vp, _vp]),
    "hp_softargmax_forward": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "hp_softargmax_backward": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "hp_bce_dice_forward": (_i, [_fp, _fp, C.c_long, C.c_float, _vp, _fp, _vp]),
    "hp_bce_dice_backward": (_i, [_fp, _fp, _vp, _fp, _fp, C.c_long, C.c_float, _vp]),
    "hp_weighted_mse_forward": (_i, [_fp, _fp, _fp, C.c_long, C.c_float, _fp, _vp]),
    "hp_weighted_mse_backward": (_i, [_fp, _fp, _fp, _fp, C.c_long, C.c_float, _fp, _vp]),
    "hp_depth_top4": (_i, [_fp, _fp, _fp, C.c_long, _i, C.c_long, _vp]),
    "hp_noise_blur_poisson": (_i, [_fp, _fp, C.c_long, _fp, _i, _i, C.c_ulonglong, _vp]),
    "hp_bce_dice_partial": (_i, [_fp, _fp, C.c_long, _vp, _vp]),
    "hp_bce_dice_finalize": (_i, [_vp, C.c_long, C.c_float, _fp, _vp]),
    "hp_bce_dice_backward_scaled": (_i, [_fp, _fp, _vp, _fp, _fp, C.c_long, C.c_float, C.c_float, _vp]),
    "hp_linear_forward": (_i, [_fp, _fp, _fp, _fp, _fp, C.c_long, _i, _i, _i, _vp]),
    "hp_linear_geglu_forward": (_i, [_fp, _fp, _fp, _fp, C.c_long, _i, _i, _i, _vp]),
    "hp_sformer_patchify": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "hp_layernorm_forward": (_i, [_fp, _fp, C.c_long, _i, _fp, _fp, C.c_float, _i, C.c_long, _vp]),
    "hp_geglu_forward": (_i, [_fp, _fp, C.c_long, _i, _vp]),
    "hp_gelu_forward": (_i, [_fp, _fp, C.c_long, _vp]),
    "hp_sformer_qkv_prepare": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, C.c_float, _fp, _fp, _i, _vp]),
    "hp_sformer_attention_workspace_bytes": (_sz, [_i, _i, _i]),
    "hp_lct_time_window": (_i, [_fp, _fp, _i, _i, _i, _i, C.c_long, C.POINTER(C.c_int), _i, _vp]),
    "hp_rgbe_decode": (_i, [_vp, _sz, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp, _sz]),
    "hp_ingest_rgbe_to_gray": (_i, [_vp, C.c_long, _fp, _fp, _vp]),
    "hp_ingest_image_to_meas": (_i, [_fp, _i, _i, _i, _i, _i, _i, _fp, _fp, _vp]),
    "hp_ingest_rgbe_to_meas": (_i, [_vp, _i, _i, _i, _i, _i, _fp, _fp, _vp]),
    "hp_box_downsample_round": (_i, [_fp, _fp, _i, _i, _i, C.c_long, C.c_long, C.c_long, _vp]),
    "hp_pair_average_axis0": (_i, [_fp, _fp, _i, _i, _i, C.c_long, C.c_long, C.c_long, _vp]),
    "hp_sformer_attention": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
}


STATS_SLOTS = 32   # HP_STATS_SLOTS (include/hiddenpose_hip.h): partial statistics vectors of hp_conv3d_forward


class ConvDesc(C.Structure):
    """Mirror of `hp_conv_desc` (include/hiddenpose_hip.h)."""
    _fields_ = [(n, C.c_int) for n in ("B", "Di", "Hi", "Wi", "Cin", "Cout", "k", "stride", "pad", "transposed", "precision", "io")]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise HiddenPoseHipError(
                        f"{LIB_PATH} is missing: the HIP extension is required (no fallback path). "
                        "Build it with `python -m hiddenpose_amd.build`.")
                # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  It must be in the
                # process BEFORE this library so that both share ONE HIP runtime; loading the
                # system runtime first and torch's second leaves the second without devices.
                import torch  # noqa: F401

                dll = C.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(dll, name)  # AttributeError here = header/library mismatch
                    fn.restype = res
                    fn.argtypes = args
                _lib = dll
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().hp_last_error_string()
        raise HiddenPoseHipError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Raw device/host address of a contiguous torch tensor or numpy array (0 for None)."""
    if t is None:
        return 0
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def current_stream_handle(device=None) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream


def profile_enable(on) -> None:
    """True / 1: every kernel family; 2: the matrix-core convolution families only; False / 0: off."""
    lib().hp_profile_enable(2 if on == 2 else 1 if on else 0)


def profile_reset() -> None:
    lib().hp_profile_reset()


def profile_read() -> dict:
    """{kernel name: (launches, total_ms)} of the HIP-event timings recorded since the last reset."""
    out = {}
    L = lib()
    for i in range(L.hp_profile_count()):
        buf = C.create_string_buffer(128)
        n = C.c_int64()
        ms = C.c_double()
        check(L.hp_profile_get(i, buf, 128, C.byref(n), C.byref(ms)), "hp_profile_get")
        out[buf.value.decode()] = (int(n.value), float(ms.value))
    return out
