"""ctypes binding of libr3det_hip.so (include/r3det_hip.h).

There is NO fallback: if the library is missing, or a tensor is not a contiguous fp32 HIP
tensor, the call raises.  PyTorch only supplies device memory and the current stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# R3DET_HIP_LIB: alternative build of the same ABI (kernel A/B experiments, tools/fr_lib_ab.py)
LIB_PATH = os.environ.get("R3DET_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libr3det_hip.so")

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> argtypes; the list doubles as the export table checked by tests/test_abi.py
SIGNATURES = {
    "r3det_rbbox_geo_mat_iou_iof": [_vp, _i, _vp, _i, _i, _vp, _vp, _sz, _vp],
    "r3det_rbbox_geo_vec_iou_iof": [_vp, _i, _vp, _i, _i, _vp, _vp],
    "r3det_box_iou_rotated_overlaps": [_vp, _i, _vp, _i, _i, _vp, _vp, _sz, _vp],
    "r3det_obb_overlaps": [_vp, _i, _vp, _i, _i, _vp, _vp, _sz, _vp],
    "r3det_box_iou_rotated_overlaps_aligned": [_vp, _vp, _i, _i, _vp, _vp],
    "r3det_mmcv_box_iou_rotated": [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _sz, _vp],
    "r3det_rbbox_assign": [_i, _vp, _i, _vp, _i, _f, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "r3det_iou_prepare_columns": [_i, _vp, _i, _vp, _sz, _vp],
    "r3det_iou_mat_prepared": [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _sz, _vp],
    "r3det_rbbox_assign_prepared": [_i, _vp, _i, _vp, _i, _vp, _f, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "r3det_iou_prepared_check": [_vp, _i, _i, _vp],
    "r3det_rbbox_assign_labeled": [_i, _vp, _i, _vp, _i, _vp, _f, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                   _vp],
    "r3det_rnms": [_vp, _vp, _i, _f, _i, _vp, _sz, _vp, _vp, _vp],
    "r3det_nms_rotated": [_vp, _vp, _i, _f, _vp, _sz, _vp, _vp, _vp],
    "r3det_ml_nms_rotated": [_vp, _vp, _vp, _i, _f, _vp, _sz, _vp, _vp, _vp],
    "r3det_mmcv_nms_rotated": [_vp, _vp, _vp, _i, _f, _vp, _sz, _vp, _vp, _vp],
    "r3det_mcnms_select": [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "r3det_mcnms_v1": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp],
    "r3det_batched_rnms": [_vp, _vp, _vp, _i, _f, _vp, _sz, _vp, _vp, _vp, _vp],
    "r3det_obb_batched_nms": [_vp, _vp, _vp, _i, _f, _vp, _sz, _vp, _vp, _vp, _vp],
    "r3det_mcnms": [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp],
    "r3det_mcnms_padded": [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _sz, _vp, _sz, _vp, _vp, _sz,
                           _vp, _vp],
    "r3det_polygon_iou": [_vp, _i, _vp, _i, _vp, _vp],
    "r3det_poly_iou_mat": [_vp, _i, _i, _vp, _i, _i, _vp, _vp],
    "r3det_nms_poly": [_vp, _vp, _i, _f, _vp, _sz, _vp, _vp, _vp],
    "r3det_convex_sort": [_vp, _vp, _i, _i, _i, _vp, _sz, _vp, _vp],
    "r3det_bias_act": [_vp, _vp, _vp, ctypes.c_longlong, _i, ctypes.c_longlong, _i, _vp],
    "r3det_filter_bboxes": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp],
    "r3det_level_pool": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _i, _i, _vp, _sz,
                         _vp],
    "r3det_levels_pool": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _f, _f, _f, _vp, _vp, _i, _vp, _sz,
                          _vp],
    "r3det_feature_refine_forward": [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _sz, _vp],
    "r3det_feature_refine_prepare": [_vp, _i, _i, _i, _f, _vp, _vp],
    "r3det_feature_refine_forward_prepared": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "r3det_frm_mix_nchw": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "r3det_feature_refine_forward_nhwc": [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp],
    "r3det_feature_refine_module_nhwc": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp],
    "r3det_feature_refine_module_prepared": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "r3det_feature_refine_backward": [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _i, _vp],
    "r3det_feature_refine_backward_ws": [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _i, _vp, _sz, _vp],
    "r3det_feature_refine_backward_index": [_vp, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp],
    "r3det_feature_refine_backward_indexed": [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _sz, _vp],
    "r3det_feature_refine_backward_nhwc": [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _i, _vp, _sz, _vp],
    "r3det_feature_refine_backward_nhwc_index": [_vp, _i, _i, _i, _f, _i, _vp, _sz, _vp],
    "r3det_feature_refine_backward_nhwc_indexed": [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _sz, _vp],
    "r3det_feature_refine_forward_levels": [_i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp],
    "r3det_feature_refine_forward_levels_nhwc": [_i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "r3det_feature_refine_module_levels_nhwc": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "r3det_feature_refine_backward_index_levels": [_i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp],
    "r3det_feature_refine_module_levels": [_i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp],
    "r3det_feature_refine_forward_levels_nhwc_tab": [_i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "r3det_feature_refine_module_levels_nhwc_tab": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp,
                                                    _vp],
    "r3det_feature_refine_backward_index_levels_tab": [_i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp],
    "r3det_feature_refine_backward_nhwc_index_levels_tab": [_i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp],
    "r3det_feature_refine_backward_nhwc_index_levels": [_i, _vp, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp],
    "r3det_feature_refine_backward_nhwc_levels_indexed": [_i, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp],
    "r3det_feature_refine_backward_levels_indexed": [_i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp],
    "r3det_set_option": [ctypes.c_char_p, _i],
    "r3det_fr_profile_read": [_vp, _i],
}

_lib = None


def lib():
    """Load the shared library once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). r3det.ops has no CPU / PyTorch fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _i
        L.r3det_nms_workspace_bytes.argtypes = [_i]
        L.r3det_nms_workspace_bytes.restype = _sz
        L.r3det_poly_nms_workspace_bytes.argtypes = [_i]
        L.r3det_poly_nms_workspace_bytes.restype = _sz
        L.r3det_rbbox_assign_workspace_bytes.argtypes = [_i, _i]
        L.r3det_rbbox_assign_workspace_bytes.restype = _sz
        L.r3det_batched_rnms_workspace_bytes.argtypes = [_i]
        L.r3det_batched_rnms_workspace_bytes.restype = _sz
        L.r3det_mcnms_workspace_bytes.argtypes = [_i, _i]
        L.r3det_mcnms_workspace_bytes.restype = _sz
        L.r3det_mcnms_select_workspace_bytes.argtypes = [_i, _i]
        L.r3det_mcnms_select_workspace_bytes.restype = _sz
        L.r3det_level_pool_workspace_bytes.argtypes = [_i, _i, _i, _i, _i]
        L.r3det_level_pool_workspace_bytes.restype = _sz
        L.r3det_levels_pool_workspace_bytes.argtypes = [_i, _i, _vp, _vp, _vp, _i]
        L.r3det_levels_pool_workspace_bytes.restype = _sz
        L.r3det_fr_table_bytes.argtypes = [_i, _i, _i]
        L.r3det_fr_table_bytes.restype = _sz
        L.r3det_fr_module_levels_workspace_bytes.argtypes = [_i, _i, _vp, _vp]
        L.r3det_fr_module_levels_workspace_bytes.restype = _sz
        L.r3det_fr_tap_table_bytes.argtypes = [_i, _i, _i]
        L.r3det_fr_tap_table_bytes.restype = _sz
        L.r3det_fr_workspace_bytes.argtypes = [_i, _i, _i, _i]
        L.r3det_fr_workspace_bytes.restype = _sz
        L.r3det_fr_levels_workspace_bytes.argtypes = [_i, _i, _vp, _vp, _i]
        L.r3det_fr_levels_workspace_bytes.restype = _sz
        L.r3det_fr_backward_levels_workspace_bytes.argtypes = [_i, _i, _vp, _vp, _i]
        L.r3det_fr_backward_levels_workspace_bytes.restype = _sz
        L.r3det_fr_backward_nhwc_levels_workspace_bytes.argtypes = [_i, _i, _vp, _vp, _i]
        L.r3det_fr_backward_nhwc_levels_workspace_bytes.restype = _sz
        L.r3det_fr_backward_workspace_bytes.argtypes = [_i, _i, _i, _i]
        L.r3det_fr_backward_workspace_bytes.restype = _sz
        L.r3det_fr_backward_nhwc_workspace_bytes.argtypes = [_i, _i, _i, _i]
        L.r3det_fr_backward_nhwc_workspace_bytes.restype = _sz
        L.r3det_iou_prepared_bytes.argtypes = [_i]
        L.r3det_iou_prepared_bytes.restype = _sz
        L.r3det_iou_workspace_bytes.argtypes = [_i, _i]
        L.r3det_iou_workspace_bytes.restype = _sz
        L.r3det_error_string.argtypes = [_i]
        L.r3det_error_string.restype = ctypes.c_char_p
        L.r3det_abi_version.restype = _i
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what}: {lib().r3det_error_string(code).decode()} ({code})")


def need_hip(t, name, dtype=torch.float32):
    """Mirror of the reference's CHECK_CUDA / CHECK_CONTIGUOUS (e.g. rbbox_geo_cuda.cpp:6-11)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor (HIP device tensor on ROCm)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def iou_workspace(n1, n2, device):
    """Scratch for the stream + drain IoU pipeline (worst-case sized, mostly untouched; the
    caching allocator makes the per-call allocation free)."""
    nbytes = int(lib().r3det_iou_workspace_bytes(n1, n2))
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def fr_profile_read(capacity=512):
    """Drain the FR cell-path profiling ring: list of (N, H, table_kernel_us, cell_kernel_us, span_us)."""
    buf = (ctypes.c_float * (5 * capacity))()
    n = lib().r3det_fr_profile_read(ctypes.cast(buf, ctypes.c_void_p), capacity)
    return [(int(buf[5 * i]), int(buf[5 * i + 1]), float(buf[5 * i + 2]), float(buf[5 * i + 3]), float(buf[5 * i + 4]))
            for i in range(n)]


def set_option(name, value):
    check(lib().r3det_set_option(name.encode(), int(value)), "r3det_set_option")
