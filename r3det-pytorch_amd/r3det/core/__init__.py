"""r3det.core -- star re-exports as in the reference (core/__init__.py:1-4; visualisation is out of
scope), so that ``from r3det.core import multiclass_nms_rotated, obb2hbb,
ranchor_inside_flags`` (models/dense_heads/rotate_anchor_head.py:11) resolves."""
from .anchor import *  # noqa: F401, F403
from .bbox import *  # noqa: F401, F403
from .post_processing import *  # noqa: F401, F403
