"""numpy front-end to the oracle (oracle/r3_oracle.cpp) and to oracle/_ref (the reference's
own CPU sources, built by oracle/build.py).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (r3det-pytorch_amd/) never imports this package.
"""
import ctypes
import os

import numpy as np

from . import build as _build

_F = ctypes.POINTER(ctypes.c_float)
_I64 = ctypes.POINTER(ctypes.c_int64)

V1, V2, V3 = 1, 2, 3
TRIG_LIBM, TRIG_TWIN = 0, 1
SORT_HOST, SORT_DEVICE = 0, 1


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_F)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build_oracle()
        _lib = ctypes.CDLL(path)
        _lib.orc_iou_mat.argtypes = [ctypes.c_int, ctypes.c_int, _F, ctypes.c_int, ctypes.c_int, _F,
                                     ctypes.c_int, ctypes.c_int, _F, ctypes.c_int]
        _lib.orc_iou_vec.argtypes = [ctypes.c_int, ctypes.c_int, _F, ctypes.c_int, ctypes.c_int, _F,
                                     ctypes.c_int, ctypes.c_int, _F]
        _lib.orc_nms.argtypes = [ctypes.c_int, _F, ctypes.c_int, _F, ctypes.c_int, ctypes.c_float,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, _I64]
        _lib.orc_nms.restype = ctypes.c_int
        _lib.orc_fr_forward.argtypes = [_F, _F] + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int,
                                                                      _F, ctypes.c_int]
        _lib.orc_fr_backward.argtypes = [_F, _F] + [ctypes.c_int] * 4 + [ctypes.c_float,
                                                                       ctypes.c_int, _F]
        _lib.orc_sincos.argtypes = [_F, ctypes.c_int, _F, _F]
    return _lib


class modes:
    """Context manager selecting the trig / hull-sort variants of the oracle."""

    def __init__(self, trig=TRIG_LIBM, sort=SORT_HOST):
        self.trig, self.sort = trig, sort

    def __enter__(self):
        L = lib()
        self.old = (L.orc_get_trig_mode(), L.orc_get_hull_sort())
        L.orc_set_trig_mode(self.trig)
        L.orc_set_hull_sort(self.sort)
        return self

    def __exit__(self, *a):
        L = lib()
        L.orc_set_trig_mode(self.old[0])
        L.orc_set_hull_sort(self.old[1])


def twin():
    """Oracle configured as the bit-exact twin of the HIP kernels."""
    return modes(TRIG_TWIN, SORT_DEVICE)


def sincos(a):
    a = _f32(a).ravel()
    s = np.empty_like(a)
    c = np.empty_like(a)
    lib().orc_sincos(_fp(a), a.size, _fp(s), _fp(c))
    return s, c


def iou_mat(geom, b1, b2, iof=False, threads=1):
    b1, b2 = _f32(b1), _f32(b2)
    n1, n2 = b1.shape[0], b2.shape[0]
    out = np.empty((n1, n2), np.float32)
    if n1 and n2:
        lib().orc_iou_mat(geom, int(iof), _fp(b1), n1, b1.shape[1], _fp(b2), n2, b2.shape[1],
                          _fp(out), threads)
    return out


def iou_vec(geom, b1, b2, iof=False):
    b1, b2 = _f32(b1), _f32(b2)
    n1, n2 = b1.shape[0], b2.shape[0]
    out = np.empty((max(n1, n2),), np.float32)
    lib().orc_iou_vec(geom, int(iof), _fp(b1), n1, b1.shape[1], _fp(b2), n2, b2.shape[1], _fp(out))
    return out


def nms(geom, boxes, scores, thr, strict=False, with_label=False, ascending=False):
    """boxes (n, 5|6) , scores (n,) -> keep (original indices)."""
    boxes, scores = _f32(boxes), _f32(scores)
    n = boxes.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    k = lib().orc_nms(geom, _fp(boxes), boxes.shape[1] if n else 5, _fp(scores), n, float(thr),
                      int(strict), int(with_label), int(ascending), keep.ctypes.data_as(_I64))
    return keep[:k].copy()


def fr_forward(feat, boxes, scale, points=1, threads=1):
    feat, boxes = _f32(feat), _f32(boxes)
    N, C, H, W = feat.shape
    out = np.empty_like(feat)
    lib().orc_fr_forward(_fp(feat), _fp(boxes), N, C, H, W, float(scale), points, _fp(out), threads)
    return out


def fr_backward(top_grad, boxes, scale, points=1):
    top_grad, boxes = _f32(top_grad), _f32(boxes)
    N, C, H, W = top_grad.shape
    out = np.zeros_like(top_grad)
    lib().orc_fr_backward(_fp(top_grad), _fp(boxes), N, C, H, W, float(scale), points, _fp(out))
    return out


# ----------------------------------------------------------------------------------------
# oracle/_ref : the reference's own CPU code
# ----------------------------------------------------------------------------------------
_ref = {}


def ref_available():
    """True when the prebuilt reference .so files exist (or can be built here)."""
    names = ["libref_v1.so", "libref_iou_v3.so", "libref_nms_v3.so", "libref_v2.so"]
    if all(os.path.exists(os.path.join(_build.OUT_REF, n)) for n in names):
        return True
    return _build.ref_available()


def _ref_lib(name):
    if name not in _ref:
        path = os.path.join(_build.OUT_REF, name)
        if not os.path.exists(path):
            _build.build_ref()
        import torch  # noqa: F401  (libtorch must be resident before the harness loads)
        _ref[name] = ctypes.CDLL(path)
    return _ref[name]


def ref_v1_iou_mat(b1, b2, iof=False):
    L = _ref_lib("libref_v1.so")
    b1, b2 = _f32(b1), _f32(b2)
    out = np.empty((b1.shape[0], b2.shape[0]), np.float32)
    L.ref_v1_iou_mat.argtypes = [_F, ctypes.c_int, ctypes.c_int, _F, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, _F]
    L.ref_v1_iou_mat(_fp(b1), b1.shape[0], b1.shape[1], _fp(b2), b2.shape[0], b2.shape[1], int(iof),
                     _fp(out))
    return out


def ref_v1_rnms(dets6, thr):
    L = _ref_lib("libref_v1.so")
    dets6 = _f32(dets6)
    n = dets6.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    L.ref_v1_rnms.argtypes = [_F, ctypes.c_int, ctypes.c_float, _I64]
    L.ref_v1_rnms.restype = ctypes.c_int
    k = L.ref_v1_rnms(_fp(dets6), n, float(thr), keep.ctypes.data_as(_I64))
    return keep[:k].copy()


def ref_v3_iou_mat(b1, b2, iof=False):
    L = _ref_lib("libref_iou_v3.so")
    b1, b2 = _f32(b1)[:, :5].copy(), _f32(b2)[:, :5].copy()
    out = np.empty((b1.shape[0], b2.shape[0]), np.float32)
    L.ref_v3_iou_mat.argtypes = [_F, ctypes.c_int, _F, ctypes.c_int, ctypes.c_int, _F]
    L.ref_v3_iou_mat(_fp(b1), b1.shape[0], _fp(b2), b2.shape[0], int(not iof), _fp(out))
    return out


def ref_v3_nms(dets5, scores, thr):
    L = _ref_lib("libref_nms_v3.so")
    dets5, scores = _f32(dets5), _f32(scores)
    n = dets5.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    L.ref_v3_nms.argtypes = [_F, _F, ctypes.c_int, ctypes.c_float, _I64]
    L.ref_v3_nms.restype = ctypes.c_int
    k = L.ref_v3_nms(_fp(dets5), _fp(scores), n, float(thr), keep.ctypes.data_as(_I64))
    return keep[:k].copy()


def ref_v2_iou_mat(b1_6, b2_6):
    L = _ref_lib("libref_v2.so")
    b1, b2 = _f32(b1_6), _f32(b2_6)
    assert b1.shape[1] == 6 and b2.shape[1] == 6
    out = np.empty((b1.shape[0], b2.shape[0]), np.float32)
    L.ref_v2_iou_mat.argtypes = [_F, ctypes.c_int, _F, ctypes.c_int, _F]
    L.ref_v2_iou_mat(_fp(b1), b1.shape[0], _fp(b2), b2.shape[0], _fp(out))
    return out


def ref_v2_nms(dets6, scores, thr):
    L = _ref_lib("libref_v2.so")
    dets6, scores = _f32(dets6), _f32(scores)
    n = dets6.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    L.ref_v2_nms.argtypes = [_F, _F, ctypes.c_int, ctypes.c_float, _I64]
    L.ref_v2_nms.restype = ctypes.c_int
    k = L.ref_v2_nms(_fp(dets6), _fp(scores), n, float(thr), keep.ctypes.data_as(_I64))
    return keep[:k].copy()


# ----------------------------------------------------------------------------------------
# "rank 4" ops: convex_sort, polygon_iou (pinned by oracle/_ref), poly_nms (parity unpinned:
# CUDA-only in the reference)
# ----------------------------------------------------------------------------------------
def polygon_iou(a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    L = lib()
    L.orc_polygon_iou.argtypes = [_F, ctypes.c_int, _F, ctypes.c_int, _F]
    L.orc_polygon_iou(_fp(a), a.shape[0], _fp(b), b.shape[0], _fp(out))
    return out


def poly_iou_mat(a, b):
    """devPolyIoU of poly_nms as a matrix; rows of a / b hold >= 8 coordinates."""
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    L = lib()
    L.orc_poly_iou_mat.argtypes = [_F, ctypes.c_int, ctypes.c_int, _F, ctypes.c_int, ctypes.c_int, _F]
    L.orc_poly_iou_mat(_fp(a), a.shape[0], a.shape[1], _fp(b), b.shape[0], b.shape[1], _fp(out))
    return out


def poly_nms(dets9, thr):
    dets9 = _f32(dets9)
    n = dets9.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    L = lib()
    L.orc_poly_nms.argtypes = [_F, ctypes.c_int, ctypes.c_float, _I64]
    L.orc_poly_nms.restype = ctypes.c_int
    k = L.orc_poly_nms(_fp(dets9), n, float(thr), keep.ctypes.data_as(_I64))
    return keep[:k].copy()


def convex_sort(pts, masks, circular=True):
    pts = _f32(pts)
    m = np.ascontiguousarray(np.asarray(masks).astype(np.float32))
    B, P = pts.shape[:2]
    out = np.empty((B, P + 1 if circular else P), np.int64)
    L = lib()
    L.orc_convex_sort.argtypes = [_F, _F, ctypes.c_int, ctypes.c_int, ctypes.c_int, _I64]
    L.orc_convex_sort(_fp(pts), _fp(m), B, P, int(circular), out.ctypes.data_as(_I64))
    return out


def ref_rank4_available():
    return all(os.path.exists(os.path.join(_build.OUT_REF, n)) for n in ("libref_convex.so", "libref_polygon.so")) \
        or _build.ref_available()


def ref_polygon_iou(a, b):
    L = _ref_lib("libref_polygon.so")
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    L.ref_polygon_iou.argtypes = [_F, ctypes.c_int, _F, ctypes.c_int, _F]
    L.ref_polygon_iou(_fp(a), a.shape[0], _fp(b), b.shape[0], _fp(out))
    return out


def ref_convex_sort(pts, masks, circular=True):
    L = _ref_lib("libref_convex.so")
    pts = _f32(pts)
    m = np.ascontiguousarray(np.asarray(masks).astype(np.uint8))
    B, P = pts.shape[:2]
    out = np.empty((B, P + 1 if circular else P), np.int64)
    L.ref_convex_sort.argtypes = [_F, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int, ctypes.c_int, ctypes.c_int, _I64]
    L.ref_convex_sort(_fp(pts), m.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), B, P, int(circular),
                      out.ctypes.data_as(_I64))
    return out
