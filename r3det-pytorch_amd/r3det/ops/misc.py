"""Exported names of the reference that sit outside the round-1 hot path."""


def convex_sort(pts, masks, circular=True):
    """Graham-scan index sort (convex/convex_wrapper.py:25-27).  Only the differentiable
    aligned obb_overlaps path uses it; listed as a later row."""
    raise NotImplementedError("convex_sort is scheduled after the hot-path rows (SURVEY.md 8f rank 4)")


def polygon_iou(poly1, poly2):
    """CPU float64 quad IoU for offline mAP (polygon_geo/polygon_geo.py:4-6): out of scope
    (offline evaluation only)."""
    raise NotImplementedError("polygon_iou (offline mAP evaluation, CPU) is out of the hot-path scope")
