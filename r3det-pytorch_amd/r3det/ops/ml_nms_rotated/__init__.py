"""r3det/ops/ml_nms_rotated/__init__.py:1."""
from ..nms import ml_nms_rotated

__all__ = ['ml_nms_rotated']
