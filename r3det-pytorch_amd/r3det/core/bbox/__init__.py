"""r3det.core.bbox -- the names the reference re-exports (core/bbox/__init__.py:1-19) that sit on or
next to the hot path.  Not restated: ``DeltaXYWHAHBBoxCoder`` (hbb heads) and ``RRandomSampler``
(sampling heads); neither is selected by the BASELINE configs."""
from .assigners import AssignResult, MaxIoUAssigner
from .coder import DeltaXYWHAOBBoxCoder
from .iou_calculators import (RBboxOverlaps2D_v1, RBboxOverlaps2D_v2, RBboxOverlaps2D_v3, rbbox_overlaps_v1,
                              rbbox_overlaps_v2, rbbox_overlaps_v3)
from .rtransforms import (hbb2obb, norm_angle, obb2hbb, obb2poly, obb2poly_np, obb2xyxy, poly2obb, poly2obb_np,
                          rbbox2result, rbbox2roi)

__all__ = [
    'RBboxOverlaps2D_v1', 'RBboxOverlaps2D_v2', 'RBboxOverlaps2D_v3',
    'rbbox_overlaps_v1', 'rbbox_overlaps_v2', 'rbbox_overlaps_v3',
    'rbbox2result', 'rbbox2roi', 'norm_angle', 'poly2obb', 'poly2obb_np',
    'obb2poly', 'obb2hbb', 'obb2xyxy', 'hbb2obb', 'obb2poly_np',
    'DeltaXYWHAOBBoxCoder', 'MaxIoUAssigner', 'AssignResult'
]
