"""r3det.ops -- same public names as the reference package (r3det/ops/__init__.py:1-15)."""
from .feature_refine import FR, FeatureRefineFunction, FeatureRefineModule, feature_refine
from .iou import box_iou_rotated, obb_overlaps, rbbox_iou
from .misc import convex_sort, polygon_iou
from .nms import (batched_rnms, ml_nms_rotated, nms_rotated, obb_batched_nms, obb_nms, poly_nms,
                  rnms)

__all__ = ['batched_rnms', 'rnms', 'rbbox_iou', 'polygon_iou',
           'FeatureRefineModule', 'obb_overlaps',
           'obb_batched_nms', 'obb_nms', 'poly_nms',
           'convex_sort', 'ml_nms_rotated']
