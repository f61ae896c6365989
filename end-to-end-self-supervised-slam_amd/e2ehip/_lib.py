"""ctypes binding of libe2eslam_hip.so (the C ABI declared in include/e2eslam.h).

There is deliberately NO fallback: if the shared library is missing or a tensor is not on a HIP
device the call raises.  The CPU restatement lives in oracle/ and is test infrastructure only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libe2eslam_hip.so")

c_fp = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float


class Strides(ctypes.Structure):
    _fields_ = [("sb", c_i64), ("sc", c_i64), ("sh", c_i64), ("sw", c_i64)]


class WgradReduceDesc(ctypes.Structure):
    """e2e_wgrad_reduce_desc (include/e2eslam.h): the slab reduction a deferred backward-weight call leaves to do."""
    _fields_ = [("slabs", c_fp), ("dw", c_fp), ("dbias", c_fp), ("scale", c_fp)] + \
               [(n, c_int) for n in ("S", "Mpad", "Npad", "Cout", "Cin", "KH", "KW", "has_bias", "accumulate", "zl")] + [("first_item", ctypes.c_longlong)]


class CopyDesc(ctypes.Structure):
    """e2e_copy_desc (include/e2eslam.h)."""
    _fields_ = [("src", c_fp), ("dst", c_fp), ("bytes", ctypes.c_longlong), ("first_item", ctypes.c_longlong)]


class E2EError(RuntimeError):
    pass


# name -> argtypes (restype int unless listed in _RESTYPE)
SIGNATURES = {
    "e2e_version": [],
    "e2e_last_error": [],
    "e2e_backproject_fwd": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_backproject_bwd": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_project3d_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_project3d_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_grid_sample_fwd": [c_fp, Strides, c_fp, c_fp] + [c_int] * 8 + [c_fp],
    "e2e_grid_sample_bwd": [c_fp, Strides, c_fp, c_fp, c_fp, c_fp] + [c_int] * 8 + [c_fp],
    "e2e_grid_sample_bwd_exact": [c_fp, Strides, c_fp, c_fp, c_fp, c_fp, c_fp] + [c_int] * 8 + [c_fp],
    "e2e_photometric_fwd": [c_fp, Strides, c_fp, Strides, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp],
    "e2e_photometric_bwd": [c_fp, Strides, c_fp, Strides, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp],
    "e2e_warp_photo_workspace_floats": [c_int, c_int, c_int],
    "e2e_warp_photo_fwd": [c_fp, c_fp, Strides, c_fp, Strides, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int,
                           c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_warp_photo_bwd": [c_fp, c_fp, Strides, c_fp, Strides, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int,
                           c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_warp_photo_lossgrad_workspace_floats": [c_int, c_int, c_int],
    "e2e_warp_photo_lossgrad": [c_fp, c_fp, Strides, c_fp, Strides, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp,
                                c_f32, c_f32, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_warp_photo_lossgrad_chain": [c_fp, c_fp, Strides, c_fp, Strides, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp,
                                      c_f32, c_f32, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_warp_photo_lossgrad_chain_flush": [c_fp, c_int, c_int, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_warp_photo_lossgrad_hostgeo": [c_fp, c_fp, Strides, c_fp, Strides, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_f32, c_f32,
                                        c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp],
    "e2e_vertex_normal_maps": [c_fp, c_fp, c_fp, c_f32, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_vertex_maps_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_transform_points": [c_fp, c_fp, c_fp, c_i64, c_int, c_fp],
    "e2e_pf_workspace_bytes": [c_i64, c_int, c_int],
    "e2e_pf_associate": [c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_fp, c_fp, c_f32, c_f32, c_fp, c_i64, c_int, c_int, c_fp],
    "e2e_pf_table": [c_int, c_i64, c_fp, c_i64, c_int, c_int, c_fp, c_fp, c_fp],
    "e2e_pf_fuse_append": [c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp],
    "e2e_pf_associate_dev": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_f32, c_f32, c_fp, c_i64, c_int, c_int, c_fp],
    "e2e_pf_fuse_append_dev": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp],
    "e2e_knn1_index_capacity_bytes": [c_i64, c_i64],
    "e2e_knn1_index_build_dev": [c_fp, c_fp, c_i64, c_i64, c_fp, c_fp],
    "e2e_knn1_index_query_dev": [c_fp, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp],
    "e2e_knn1_index_query_dev_image": [c_fp, c_i64, c_int, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp],
    "e2e_knn1_index_query_dev_image_warm": [c_fp, c_i64, c_int, c_fp, c_fp, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp],
    "e2e_knn1_index_capacity_bytes_res": [c_i64, c_i64, c_int],
    "e2e_knn1_index_build_dev_res": [c_fp, c_fp, c_i64, c_i64, c_fp, c_int, c_fp],
    "e2e_knn1_index_query_dev_res": [c_fp, c_i64, c_fp, c_fp, c_i64, c_i64, c_fp, c_int, c_fp, c_fp, c_fp],
    "e2e_knn1_workspace_bytes": [c_i64, c_i64],
    "e2e_knn1_fwd": [c_fp, c_i64, c_fp, c_i64, c_fp, c_fp, c_fp, c_int, c_fp],
    "e2e_knn1_index_build": [c_fp, c_i64, c_i64, c_fp, c_fp],
    "e2e_knn1_index_query": [c_fp, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp],
    "e2e_knn1_bwd": [c_fp, c_fp, c_fp, c_fp, c_i64, c_fp, c_fp],
    "e2e_knn1_bwd_ref": [c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_fp, c_fp, c_fp],
    "e2e_median_workspace_bytes": [],
    "e2e_median_lower": [c_fp, c_i64, c_fp, c_fp, c_fp],
    "e2e_depth_scale_workspace_bytes": [],
    "e2e_depth_scale_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_fp],
    "e2e_depth_scale_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_fp],
    "e2e_depth_scale_bwd_at": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_i64, c_fp],
    "e2e_depth_fixed_scale_fwd": [c_fp, c_f32, c_fp, c_fp, c_i64, c_fp],
    "e2e_depth_fixed_scale_bwd": [c_fp, c_fp, c_f32, c_fp, c_i64, c_fp],
    "e2e_reduce_workspace_floats": [],
    "e2e_mean_diff_fwd": [c_fp, c_fp, c_i64, c_int, c_fp, c_fp, c_fp],
    "e2e_mean_diff_bwd": [c_fp, c_fp, c_fp, c_i64, c_int, c_fp, c_fp],
    "e2e_depth_metrics": [c_fp, c_fp, c_i64, c_int, c_fp, c_fp, c_fp],
    "e2e_adam_step": [c_fp, c_fp, c_fp, c_fp, c_i64, c_f32, c_f32, c_f32, c_f32, c_int, c_fp],
    "e2e_adam_step_mean": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_f32, c_f32, c_f32, c_f32, c_int, c_fp],
    "e2e_adam_step_resident": [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_f32, c_f32, c_f32, c_fp, c_int, c_fp, c_fp],
    "e2e_mask_mul": [c_fp, Strides, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "e2e_channel_mean": [c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "e2e_mean_normalize": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "e2e_masked_mean_lossgrad": [c_fp, c_fp, c_i64, c_f32, c_fp, c_fp, c_fp, c_fp],
    "e2e_aux_workspace_floats": [],
    "e2e_smoothness_lossgrad": [c_fp, c_fp, Strides, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp],
    "e2e_geometric_consistency_lossgrad": [c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp],
    "e2e_masked_l1_lossgrad": [c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_fp, c_fp],
    "e2e_min_reprojection_lossgrad": [c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp],
    "e2e_disp_blend_fwd": [c_fp, c_int, c_int, c_fp, c_fp],
    "e2e_disp_blend_bwd": [c_fp, c_int, c_int, c_fp, c_fp],
    "e2e_conv_weight_layouts": [c_fp, c_int, c_int, c_int, c_int, c_fp, c_int, c_fp, c_int, c_fp],
    "e2e_conv_weight_layouts_batched": [c_fp, c_int, c_fp],
    "e2e_conv2d_fwd": [c_fp, c_fp, c_int, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_fp] + [c_int] * 11 + [c_f32, c_f32, c_fp, c_fp],
    "e2e_conv2d_splitk_workspace_floats": [c_i64, c_int, c_int],
    "e2e_conv2d_bwd_data_workspace_floats": [c_int, c_int, c_int, c_int, c_int, c_int],
    "e2e_conv2d_fwd_tuned": [c_fp, c_fp, c_int, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_fp] + [c_int] * 11 + [c_f32, c_f32, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_conv2d_bwd_data_fused_tuned": [c_fp, c_fp, c_int, c_fp] + [c_int] * 13 + [c_fp, c_int, c_fp, c_fp, c_int, c_int, c_int, c_fp],
    "e2e_conv2d_bwd_weight_scaled_tuned": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp] + [c_int] * 13 + [c_f32, c_f32, c_int, c_fp],
    "e2e_conv2d_wgrad_tuned_workspace_floats": [c_int] * 9,
    "e2e_conv_tuned_workspace_floats": [c_i64, c_int],
    "e2e_conv_workspace_flag_floats": [],
    "e2e_conv_streamk_error_index": [],
    "e2e_conv_gemm_choice": [c_i64, c_int, c_int, c_int, c_int, c_fp],
    "e2e_conv2d_act_bwd": [c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_int, c_fp],
    "e2e_conv2d_bwd_data": [c_fp, c_fp, c_int, c_fp] + [c_int] * 12 + [c_fp, c_fp],
    "e2e_conv2d_bwd_data_acc": [c_fp, c_fp, c_int, c_fp] + [c_int] * 13 + [c_fp, c_fp],
    "e2e_conv2d_bwd_data_fused": [c_fp, c_fp, c_int, c_fp] + [c_int] * 13 + [c_fp, c_int, c_fp, c_fp, c_fp],
    "e2e_conv2d_gather_adjoint_act": [c_fp] + [c_int] * 7 + [c_fp, c_fp, c_int, c_int, c_fp, c_int, c_fp, c_int, c_fp],
    "e2e_conv2d_bwd_weight_scaled": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp] + [c_int] * 13 + [c_f32, c_f32, c_fp],
    "e2e_conv2d_bwd_weight_scaled_deferred": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp] + [c_int] * 13 + [c_f32, c_f32, ctypes.POINTER(WgradReduceDesc), c_fp],
    "e2e_wgrad_reduce_batch_prepare": [ctypes.POINTER(WgradReduceDesc), c_int],
    "e2e_wgrad_reduce_batched": [c_fp, c_int, ctypes.c_longlong, c_fp],
    "e2e_copy_batch_prepare": [ctypes.POINTER(CopyDesc), c_int],
    "e2e_copy_batched": [c_fp, c_int, ctypes.c_longlong, c_fp],
    "e2e_head_bwd_act": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp],
    "e2e_conv2d_act_bwd_acc": [c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_int, c_int, c_fp],
    "e2e_conv2d_gather_adjoint": [c_fp] + [c_int] * 7 + [c_fp, c_fp, c_int, c_int, c_fp],
    "e2e_conv2d_wgrad_workspace_floats": [c_int] * 8,
    "e2e_conv2d_bwd_weight": [c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp] + [c_int] * 13 + [c_f32, c_f32, c_fp],
    "e2e_maxpool3x3s2_fwd": [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp],
    "e2e_maxpool3x3s2_bwd": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp],
    "e2e_maxpool3x3s2_fwd_idx": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp],
    "e2e_maxpool3x3s2_bwd_idx": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp],
    "e2e_bn_fold": [c_fp, c_fp, c_fp, c_fp, c_f32, c_fp, c_fp, c_fp, c_int, c_fp],
    "e2e_affine_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_i64, c_int, c_fp],
    "e2e_affine_bwd_workspace_floats": [c_int],
    "e2e_affine_bwd": [c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_fp, c_fp, c_int, c_fp, c_fp],
    "e2e_upsample2_concat": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp],
    "e2e_head_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp],
    "e2e_head_workspace_floats": [],
    "e2e_head_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp],
    "e2e_icp_workspace_bytes": [],
    "e2e_icp_state_doubles": [],
    "e2e_icp_state_init": [c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_double, c_fp],
    "e2e_icp_update": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_fp],
    "e2e_icp_reduce_update": [c_fp, c_fp, c_fp, c_fp, c_fp, c_f32, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int,
                              ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_fp],
    "e2e_icp_source_subsample": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "e2e_pf_active_subsample_dev": [c_fp, c_fp, c_fp, c_i64, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_i64, c_fp],
    "e2e_icp_normal_equations": [c_fp, c_fp, c_fp, c_fp, c_fp, c_f32, c_i64, c_fp, c_fp, c_fp],
}
_RESTYPE = {"e2e_last_error": ctypes.c_char_p, "e2e_warp_photo_workspace_floats": c_i64,
            "e2e_warp_photo_lossgrad_workspace_floats": c_i64, "e2e_pf_workspace_bytes": c_i64,
            "e2e_knn1_workspace_bytes": c_i64, "e2e_knn1_index_capacity_bytes": c_i64, "e2e_knn1_index_capacity_bytes_res": c_i64, "e2e_median_workspace_bytes": c_i64,
            "e2e_depth_scale_workspace_bytes": c_i64, "e2e_reduce_workspace_floats": c_i64,
            "e2e_conv2d_wgrad_workspace_floats": c_i64, "e2e_conv2d_wgrad_tuned_workspace_floats": c_i64, "e2e_conv_tuned_workspace_floats": c_i64, "e2e_conv2d_splitk_workspace_floats": c_i64, "e2e_conv2d_bwd_data_workspace_floats": c_i64,
            "e2e_head_workspace_floats": c_i64, "e2e_icp_workspace_bytes": c_i64, "e2e_icp_state_doubles": c_i64, "e2e_aux_workspace_floats": c_i64,
            "e2e_affine_bwd_workspace_floats": c_i64, "e2e_wgrad_reduce_batch_prepare": ctypes.c_longlong, "e2e_copy_batch_prepare": ctypes.c_longlong}

_lib = None


def load():
    """Load the shared library (once).  Raises E2EError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise E2EError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
                           "(there is no CPU / PyTorch fallback for the hot path)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, c_int)
        _lib = lib
    return _lib


PROFILE_HOOK = [None]           # e2ehip.profile.KernelTimer while a timing pass is active


def call(name, *args):
    lib = load()
    hook = PROFILE_HOOK[0]
    rc = getattr(lib, name)(*args) if hook is None else hook.around(name, args, lambda: getattr(lib, name)(*args))
    if rc != 0:
        raise E2EError(f"{name} failed ({rc}): {lib.e2e_last_error().decode()}")


def stream():
    return torch.cuda.current_stream().cuda_stream


def dev(t, name="tensor", dtype=torch.float32):
    """Validate a tensor for the HIP path and return it."""
    if not torch.is_tensor(t):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise E2EError(f"{name}: the e2eslam hot path runs on the HIP device only (got a {t.device} tensor); "
                       "there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def strides4(t):
    """Element strides of a (B,C,H,W)-indexed view."""
    s = t.stride()
    return Strides(s[0], s[1], s[2], s[3])
