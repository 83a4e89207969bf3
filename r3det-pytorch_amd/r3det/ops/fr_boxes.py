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
