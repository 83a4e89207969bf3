from .box_iou_rotated_wrapper import obb_overlaps  # noqa: F401, F403
