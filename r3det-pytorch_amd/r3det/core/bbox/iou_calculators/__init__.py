from .rotate_iou2d_calculator import (RBboxOverlaps2D_v1, RBboxOverlaps2D_v2, RBboxOverlaps2D_v3,
                                      rbbox_overlaps_v1, rbbox_overlaps_v2, rbbox_overlaps_v3)

__all__ = ['RBboxOverlaps2D_v1', 'RBboxOverlaps2D_v2', 'RBboxOverlaps2D_v3',
           'rbbox_overlaps_v1', 'rbbox_overlaps_v2', 'rbbox_overlaps_v3']
