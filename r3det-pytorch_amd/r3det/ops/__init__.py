"""r3det.ops -- same public names and per-op subpackages as the reference package
(r3det/ops/__init__.py:1-15).  The arithmetic lives in the flat modules (iou.py, nms.py,
feature_refine.py, misc.py) over libr3det_hip.so; the subpackages carry the reference's module
names (``r3det.ops.rnms.rnms_wrapper`` ...) so that its import lines resolve unchanged."""
from .box_iou_rotated import obb_overlaps
from .convex import convex_sort
from .fr import FeatureRefineModule
from .ml_nms_rotated import ml_nms_rotated
from .nms_rotated import obb_batched_nms, obb_nms, poly_nms
from .polygon_geo import polygon_iou
from .rbbox_geo import rbbox_iou
from .rnms import batched_rnms, rnms

__all__ = ['batched_rnms', 'rnms', 'rbbox_iou', 'polygon_iou',
           'FeatureRefineModule', 'obb_overlaps',
           'obb_batched_nms', 'obb_nms', 'poly_nms',
           'convex_sort', 'ml_nms_rotated']
