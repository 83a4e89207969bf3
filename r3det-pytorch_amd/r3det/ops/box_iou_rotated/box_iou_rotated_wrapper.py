"""r3det/ops/box_iou_rotated/box_iou_rotated_wrapper.py:8-216 under its module name."""
from ..iou import (aligned_obb_overlaps, convex_areas, obb2poly, obb_overlaps,  # noqa: F401
                   poly_intersection, shoelace)
