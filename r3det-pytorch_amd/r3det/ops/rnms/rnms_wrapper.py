"""r3det/ops/rnms/rnms_wrapper.py:7-69 under its module name."""
from ..nms import batched_rnms, rnms  # noqa: F401
