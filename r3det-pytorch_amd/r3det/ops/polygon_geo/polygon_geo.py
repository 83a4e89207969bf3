"""r3det/ops/polygon_geo/polygon_geo.py:4-6 under its module name."""
from ..misc import polygon_iou  # noqa: F401
