"""IoU calculators selected by config string (``iou_calculator=dict(type='RBboxOverlaps2D_v1')``).

Mirror of r3det/core/bbox/iou_calculators/rotate_iou2d_calculator.py:
  RBboxOverlaps2D_v1 :7-43   -> rbbox_overlaps_v1 :51-80   -> r3det.ops.rbbox_iou        (v1)
  RBboxOverlaps2D_v2 :83-124 -> rbbox_overlaps_v2 :127-156 -> mmcv-style box_iou_rotated (v2)
  RBboxOverlaps2D_v3 :159-199-> rbbox_overlaps_v3 :202-231 -> r3det.ops.obb_overlaps     (v3)
"""
from ....ops import obb_overlaps, rbbox_iou
from ....ops.mmcv_ops import box_iou_rotated
from ....registry import IOU_CALCULATORS


def _check(bboxes1, bboxes2, mode, is_aligned):
    assert mode in ['iou', 'iof']
    assert bboxes1.size(-1) == 5 or bboxes1.size(0) == 0
    assert bboxes2.size(-1) == 5 or bboxes2.size(0) == 0
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if is_aligned:
        assert rows == cols
    return rows, cols


def _empty(bboxes1, rows, cols, is_aligned):
    # uninitialised on purpose, like ``bboxes1.new(rows, cols)`` (:77-78)
    return bboxes1.new_empty((rows, 1)) if is_aligned else bboxes1.new_empty((rows, cols))


def rbbox_overlaps_v1(bboxes1, bboxes2, mode='iou', is_aligned=False):
    rows, cols = _check(bboxes1, bboxes2, mode, is_aligned)
    if rows * cols == 0:
        return _empty(bboxes1, rows, cols, is_aligned)
    return rbbox_iou(bboxes1, bboxes2, is_aligned, mode == 'iof')


def rbbox_overlaps_v2(bboxes1, bboxes2, mode='iou', is_aligned=False):
    rows, cols = _check(bboxes1, bboxes2, mode, is_aligned)
    if rows * cols == 0:
        return _empty(bboxes1, rows, cols, is_aligned)
    return box_iou_rotated(bboxes1, bboxes2, mode, is_aligned)


def rbbox_overlaps_v3(bboxes1, bboxes2, mode='iou', is_aligned=False):
    rows, cols = _check(bboxes1, bboxes2, mode, is_aligned)
    if rows * cols == 0:
        return _empty(bboxes1, rows, cols, is_aligned)
    return obb_overlaps(bboxes1, bboxes2, mode, is_aligned)


class _RBboxOverlaps2D:
    """Common call protocol (:11-43): accepts (m,5) or (m,6) [score column dropped]."""
    _fn = None
    _contiguous = True

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False, version='v1'):
        assert bboxes1.size(-1) in [0, 5, 6]
        assert bboxes2.size(-1) in [0, 5, 6]
        if bboxes2.size(-1) == 6:
            bboxes2 = bboxes2[..., :5]
        if bboxes1.size(-1) == 6:
            bboxes1 = bboxes1[..., :5]
        if self._contiguous:
            bboxes1, bboxes2 = bboxes1.contiguous(), bboxes2.contiguous()
        return type(self)._fn(bboxes1, bboxes2, mode, is_aligned)

    def __repr__(self):
        return self.__class__.__name__ + '()'


@IOU_CALCULATORS.register_module()
class RBboxOverlaps2D_v1(_RBboxOverlaps2D):
    _fn = staticmethod(rbbox_overlaps_v1)


@IOU_CALCULATORS.register_module()
class RBboxOverlaps2D_v2(_RBboxOverlaps2D):
    _fn = staticmethod(rbbox_overlaps_v2)


@IOU_CALCULATORS.register_module()
class RBboxOverlaps2D_v3(_RBboxOverlaps2D):
    _fn = staticmethod(rbbox_overlaps_v3)
    _contiguous = False  # v3 passes views through; obb_overlaps makes them contiguous (:196)
