from .bbox_nms_rotated import multiclass_nms_rotated

__all__ = ['multiclass_nms_rotated']
