"""Stand-ins for the two third-party operators the reference imports from mmcv-full 1.3.15..1.5.0
(``from mmcv.ops import box_iou_rotated`` rotate_iou2d_calculator.py:1, ``from mmcv.ops import
nms_rotated`` bbox_nms_rotated.py:2), under mmcv's own names.  mmcv is not under the reference tree:
the geometry follows the in-tree statement of the same convention (ops/ml_nms_rotated/src), parity
is UNPINNED."""
from .iou import box_iou_rotated  # noqa: F401
from .nms import nms_rotated  # noqa: F401

__all__ = ['box_iou_rotated', 'nms_rotated']
