"""r3det/ops/rbbox_geo/rbbox_geo.py:4-9 -- ``rbbox_iou(rb1, rb2, vec=False, iof=False)``."""
from ..iou import rbbox_iou  # noqa: F401
