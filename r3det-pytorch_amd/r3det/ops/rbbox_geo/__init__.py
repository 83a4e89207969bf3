from .rbbox_geo import rbbox_iou

__all__ = ['rbbox_iou']
