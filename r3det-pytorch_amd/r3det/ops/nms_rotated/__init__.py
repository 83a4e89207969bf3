from .nms_rotated_wrapper import obb_batched_nms, obb_nms, poly_nms

__all__ = ['obb_batched_nms', 'obb_nms', 'poly_nms']
