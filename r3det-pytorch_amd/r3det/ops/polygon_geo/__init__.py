from .polygon_geo import polygon_iou

__all__ = ['polygon_iou']
