"""r3det/ops/convex/convex_wrapper.py:8-27 under its module name."""
from ..misc import convex_sort  # noqa: F401
