"""Fused producers of the Feature Refinement boxes (include/r3det_hip.h: r3det_filter_bboxes).

``filter_bboxes`` = RRetinaHead.filter_bboxes of one level for a whole batch
(models/dense_heads/rotate_retina_head.py:117-179); ``refine_bboxes`` = RRetinaRefineHead.refine_bboxes
(rotate_retina_refine_head.py:56-97).  The head outputs are read in place through their strides
(NCHW or channels_last); the result is the (N, H*W, 5) array the FR sampler reads.
"""
import ctypes
import math

import torch

from .. import _C

MAX_RATIO = abs(math.log(16 / 1000))  # delta2bbox_v1: wh_ratio_clip = 16 / 1000
POOL_MAX_K = 4096          # r3det_level_pool: largest top-k (the select kernel's LDS sort)
POOL_MAX_ROWS = 1_000_000  # ... and most rows per image and level when a top-k is needed (csrc/r3_pool.hip)


def _strides(t):
    return (ctypes.c_longlong * 4)(*t.stride())


def _check(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        raise RuntimeError(f"{name} must be a 4-d fp32 CUDA tensor")
    return t


def filter_bboxes(cls_score, bbox_pred, anchors, num_anchors, num_classes):
    """cls_score (N, A*C, H, W), bbox_pred (N, A*5, H, W), anchors (H*W*A, 5) -> (N, H*W, 5)."""
    cls_score, bbox_pred = _check(cls_score, "cls_score"), _check(bbox_pred, "bbox_pred")
    N, _, H, W = cls_score.shape
    assert cls_score.size(1) == num_anchors * num_classes and bbox_pred.size(1) == num_anchors * 5
    anchors = _C.need_hip(anchors.contiguous(), "anchors")
    assert anchors.shape == (H * W * num_anchors, 5)
    out = torch.empty((N, H * W, 5), dtype=torch.float32, device=cls_score.device)
    with torch.cuda.device(cls_score.device):
        _C.check(_C.lib().r3det_filter_bboxes(_C.ptr(cls_score), _strides(cls_score), _C.ptr(bbox_pred),
                                              _strides(bbox_pred), _C.ptr(anchors), 0, N, num_anchors, num_classes,
                                              H, W, MAX_RATIO, _C.ptr(out), _C.stream()), "r3det_filter_bboxes")
    return out


def refine_bboxes(bbox_pred, rois):
    """bbox_pred (N, 5, H, W), rois (N, H*W, 5) -> delta2bbox_v1(rois, pred) (N, H*W, 5)."""
    bbox_pred = _check(bbox_pred, "bbox_pred")
    N, _, H, W = bbox_pred.shape
    rois = _C.need_hip(rois.contiguous(), "rois")
    assert rois.shape == (N, H * W, 5) and bbox_pred.size(1) == 5
    out = torch.empty((N, H * W, 5), dtype=torch.float32, device=bbox_pred.device)
    with torch.cuda.device(bbox_pred.device):
        _C.check(_C.lib().r3det_filter_bboxes(None, None, _C.ptr(bbox_pred), _strides(bbox_pred), _C.ptr(rois), 1, N,
                                              1, 1, H, W, MAX_RATIO, _C.ptr(out), _C.stream()), "r3det_filter_bboxes")
    return out


def level_pool(cls_score, bbox_pred, anchors, num_anchors, num_classes, nms_pre, max_shape, pool_boxes, pool_scores,
               row_offset):
    """One pyramid level of the pre-NMS pool for a whole batch (r3det_level_pool): sigmoid, per-image top
    ``nms_pre`` rows by the best class score (score order), delta2bbox_v1 with the centre clamp to ``max_shape``
    -- written into rows [row_offset, row_offset + min(nms_pre, H*W*A)) of ``pool_boxes`` (N, n, 5) and
    ``pool_scores`` (N, n, C + 1).  ``anchors``: (H*W*A, 5), or (N, H*W*A, 5) per image (refine head).
    Returns the number of rows written per image."""
    cls_score, bbox_pred = _check(cls_score, "cls_score"), _check(bbox_pred, "bbox_pred")
    N, _, H, W = cls_score.shape
    A, C = num_anchors, num_classes
    assert cls_score.size(1) == A * C and bbox_pred.size(1) == A * 5
    per_image = anchors.dim() == 3
    anchors = _C.need_hip(anchors.contiguous(), "anchors")
    L = H * W * A
    assert anchors.shape == ((N, L, 5) if per_image else (L, 5))
    _C.need_hip(pool_boxes, "pool_boxes")
    _C.need_hip(pool_scores, "pool_scores")
    n = pool_boxes.size(1)
    assert pool_boxes.shape == (N, n, 5) and pool_scores.shape == (N, n, C + 1)
    lib = _C.lib()
    k = int(nms_pre) if nms_pre is not None else -1
    with torch.cuda.device(cls_score.device):
        wsb = int(lib.r3det_level_pool_workspace_bytes(N, A, H, W, k))
        ws = torch.empty(wsb, dtype=torch.uint8, device=cls_score.device)
        mx, my = (float(max_shape[1] - 1), float(max_shape[0] - 1)) if max_shape is not None else (-1.0, -1.0)
        _C.check(lib.r3det_level_pool(_C.ptr(cls_score), _strides(cls_score), _C.ptr(bbox_pred), _strides(bbox_pred),
                                      _C.ptr(anchors), int(per_image), N, A, C, H, W, k, MAX_RATIO, mx, my,
                                      _C.ptr(pool_boxes), _C.ptr(pool_scores), n, int(row_offset), _C.ptr(ws), wsb,
                                      _C.stream()), "r3det_level_pool")
    return k if 0 < k < L else L


POOL_MAX_LEVELS = 8  # r3det_levels_pool: levels per call


def levels_pool(cls_scores, bbox_preds, anchors, num_anchors, num_classes, nms_pre, max_shape, pool_boxes, pool_scores):
    """All pyramid levels of the pre-NMS pool in one library call (r3det_levels_pool): ``cls_scores`` /
    ``bbox_preds`` / ``anchors`` are per-level lists as for ``level_pool``; level l's rows follow level l - 1's,
    starting at row 0 of the pool arrays.  Row for row what one ``level_pool`` call per level writes.  Returns the
    number of rows written per image."""
    nl = len(cls_scores)
    assert 0 < nl <= POOL_MAX_LEVELS and len(bbox_preds) == nl and len(anchors) == nl
    A, C = num_anchors, num_classes
    N = cls_scores[0].size(0)
    per_image = anchors[0].dim() == 3
    anchors = [_C.need_hip(a.contiguous(), "anchors") for a in anchors]
    Hs, Ws, rows = [], [], 0
    k = int(nms_pre) if nms_pre is not None else -1
    for cls, reg, anc in zip(cls_scores, bbox_preds, anchors):
        _check(cls, "cls_score"), _check(reg, "bbox_pred")
        n_, _, H, W = cls.shape
        L = H * W * A
        assert n_ == N and cls.size(1) == A * C and reg.shape == (N, A * 5, H, W)
        assert (anc.dim() == 3) == per_image and anc.shape == ((N, L, 5) if per_image else (L, 5))
        Hs.append(H), Ws.append(W)
        rows += k if 0 < k < L else L
    _C.need_hip(pool_boxes, "pool_boxes")
    _C.need_hip(pool_scores, "pool_scores")
    n = pool_boxes.size(1)
    assert pool_boxes.shape == (N, n, 5) and pool_scores.shape == (N, n, C + 1) and rows <= n
    ptrs = lambda ts: (ctypes.c_void_p * nl)(*[t.data_ptr() for t in ts])  # noqa: E731
    strides = lambda ts: (ctypes.c_longlong * (4 * nl))(*[v for t in ts for v in t.stride()])  # noqa: E731
    ints = lambda vs: (ctypes.c_int * nl)(*vs)  # noqa: E731
    lib = _C.lib()
    dev = cls_scores[0].device
    with torch.cuda.device(dev):
        cA, cH, cW = ints([A] * nl), ints(Hs), ints(Ws)
        wsb = int(lib.r3det_levels_pool_workspace_bytes(nl, N, cA, cH, cW, k))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        mx, my = (float(max_shape[1] - 1), float(max_shape[0] - 1)) if max_shape is not None else (-1.0, -1.0)
        _C.check(lib.r3det_levels_pool(nl, ptrs(cls_scores), strides(cls_scores), ptrs(bbox_preds), strides(bbox_preds),
                                       ptrs(anchors), int(per_image), N, cA, C, cH, cW, k, MAX_RATIO, mx, my,
                                       _C.ptr(pool_boxes), _C.ptr(pool_scores), n, _C.ptr(ws), wsb, _C.stream()),
                 "r3det_levels_pool")
    return rows
