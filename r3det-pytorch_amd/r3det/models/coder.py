"""Kept for the model glue: the coder lives where the reference has it
(core/bbox/coder/delta_xywha_rbbox_coder.py)."""
from ..core.bbox.coder.delta_xywha_rbbox_coder import bbox2delta_v1, delta2bbox_v1  # noqa: F401
