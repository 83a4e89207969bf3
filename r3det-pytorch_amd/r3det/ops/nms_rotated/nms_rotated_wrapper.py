"""r3det/ops/nms_rotated/nms_rotated_wrapper.py:7-98 under its module name."""
from ..nms import obb2hbb, obb_batched_nms, obb_nms, poly_nms  # noqa: F401
