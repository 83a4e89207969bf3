from .bbox_nms_rotated import CapacityHint, PaddedNms, multiclass_nms_rotated, multiclass_nms_rotated_batch

__all__ = ['multiclass_nms_rotated', 'multiclass_nms_rotated_batch', 'CapacityHint', 'PaddedNms']
