"""Rotated-box IoU operators.

Host-side mirror of the reference wrappers
  r3det/ops/rbbox_geo/rbbox_geo.py:4-9                     (rbbox_iou, geometry v1)
  r3det/ops/box_iou_rotated/box_iou_rotated_wrapper.py:8-64 (obb_overlaps, geometry v3)
  mmcv.ops.box_iou_rotated (call site rotate_iou2d_calculator.py:156, geometry v2)
over the C ABI in include/r3det_hip.h.  Same names, argument meaning and error behaviour;
the arithmetic runs in libr3det_hip.so only.
"""
import numpy as np
import torch

from .. import _C


def _as_boxes(t, name):
    t = _C.need_hip(t, name)
    if t.dim() != 2 or (t.numel() > 0 and t.size(1) != 5):
        raise RuntimeError(f"{name} must have shape (n, 5), got {tuple(t.shape)}")
    return t


GEOM = {'v1': 1, 'v2': 2, 'v3': 3}


def prepare_columns(boxes, version='v1'):
    """What the overlap kernels need of a box list that is the SECOND operand of many matrices / assignments -- the
    anchor grid of a training run -- computed once (r3det_iou_prepare_columns): exact per-box records, the data of the
    conservative disjointness test, the bounding box of every 256 consecutive boxes.  Returns an opaque device buffer
    for ``rbbox_iou(..., prepared=)`` / ``MaxIoUAssigner.assign(..., shared_key=)``; valid for exactly these boxes
    and this geometry."""
    b = _as_boxes(boxes, "boxes")
    n = b.size(0)
    L = _C.lib()
    with torch.cuda.device(b.device):
        nbytes = int(L.r3det_iou_prepared_bytes(n))
        buf = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
        if n:
            _C.check(L.r3det_iou_prepare_columns(GEOM[version], _C.ptr(b), n, _C.ptr(buf), nbytes, _C.stream()),
                     "iou_prepare_columns")
    # (what it was prepared for rides on the tensor object: the wrappers refuse another geometry / column count on
    # the host, without a device read; the buffer's own header is what the kernels check)
    buf.r3_prepared_for = (GEOM[version], n)
    return buf


def _check_prepared(prepared, geom, n):
    want = getattr(prepared, "r3_prepared_for", None)
    if want is not None and want != (geom, n):
        raise ValueError(f"prepared columns were built for (geometry, n) = {want}, this call has {(geom, n)}")


def rbbox_iou(rb1, rb2, vec=False, iof=False, prepared=None):
    """IoU (or IoF) of rotated boxes with the v1 vertex/segment geometry.

    ``vec=False`` -> (n1, n2) matrix (mat_iou_iof); ``vec=True`` -> (max(n1, n2),) with modulo
    broadcast (vec_iou_iof).  Inputs must be contiguous fp32 device tensors, as the
    reference's CHECK_INPUT demands (rbbox_geo_cuda.cpp:6-18).  ``prepared``: ``prepare_columns(rb2, 'v1')``
    (matrix form only; same results).
    """
    rb1, rb2 = _as_boxes(rb1, "rb1"), _as_boxes(rb2, "rb2")
    n1, n2 = rb1.size(0), rb2.size(0)
    L = _C.lib()
    with torch.cuda.device(rb1.device):
        if vec:
            out = rb1.new_empty((max(n1, n2),))
            if n1 and n2:
                _C.check(L.r3det_rbbox_geo_vec_iou_iof(_C.ptr(rb1), n1, _C.ptr(rb2), n2, int(bool(iof)),
                                                       _C.ptr(out), _C.stream()), "vec_iou_iof")
        else:
            out = rb1.new_empty((n1, n2))
            if n1 and n2:
                ws, wsb = _C.iou_workspace(n1, n2, rb1.device)
                if prepared is not None:
                    _check_prepared(prepared, 1, n2)
                    _C.check(L.r3det_iou_mat_prepared(1, _C.ptr(rb1), n1, _C.ptr(rb2), n2, _C.ptr(prepared),
                                                      int(bool(iof)), _C.ptr(out), _C.ptr(ws), wsb, _C.stream()),
                             "iou_mat_prepared")
                else:
                    _C.check(L.r3det_rbbox_geo_mat_iou_iof(_C.ptr(rb1), n1, _C.ptr(rb2), n2, int(bool(iof)),
                                                           _C.ptr(out), _C.ptr(ws), wsb, _C.stream()), "mat_iou_iof")
    return out


def box_iou_rotated_v3(b1, b2, iou=True):
    """box_iou_rotated_ext.overlaps(b1, b2, iou_or_iof) (box_iou_rotated_ext.cpp:17-32)."""
    b1 = _as_boxes(b1.contiguous(), "boxes1")
    b2 = _as_boxes(b2.contiguous(), "boxes2")
    n1, n2 = b1.size(0), b2.size(0)
    out = b1.new_empty((n1, n2))
    if n1 and n2:
        with torch.cuda.device(b1.device):
            ws, wsb = _C.iou_workspace(n1, n2, b1.device)
            _C.check(_C.lib().r3det_box_iou_rotated_overlaps(_C.ptr(b1), n1, _C.ptr(b2), n2, int(bool(iou)),
                                                             _C.ptr(out), _C.ptr(ws), wsb, _C.stream()), "overlaps")
    return out


def box_iou_rotated(bboxes1, bboxes2, mode='iou', aligned=False):
    """Stand-in for ``mmcv.ops.box_iou_rotated`` (v2 / standard vertex sign).

    mmcv is not part of the reference tree; the geometry follows the in-tree statement of the
    same convention, ml_nms_rotated/src/box_iou_rotated_utils.h.
    """
    assert mode in ['iou', 'iof']
    b1 = _as_boxes(bboxes1.contiguous(), "bboxes1")
    b2 = _as_boxes(bboxes2.contiguous(), "bboxes2")
    n1, n2 = b1.size(0), b2.size(0)
    if aligned:
        assert n1 == n2
        out = b1.new_empty((n1,))
    else:
        out = b1.new_empty((n1, n2))
    if n1 and n2:
        with torch.cuda.device(b1.device):
            ws, wsb = (None, 0) if aligned else _C.iou_workspace(n1, n2, b1.device)
            _C.check(_C.lib().r3det_mmcv_box_iou_rotated(_C.ptr(b1), n1, _C.ptr(b2), n2,
                                                         0 if mode == 'iou' else 1, int(bool(aligned)),
                                                         _C.ptr(out), _C.ptr(ws) if ws is not None else None, wsb,
                                                         _C.stream()), "box_iou_rotated")
    return out


def _to_device_tensor(a, device_id):
    dev = torch.device('cuda', torch.cuda.current_device() if device_id is None else device_id)
    return torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)


def obb_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, device_id=None):
    """Overlaps of oriented boxes, geometry v3 (box_iou_rotated_wrapper.py:8-64).

    Tensor or numpy inputs; numpy goes to ``cuda:device_id`` (current device when None --
    this build has no CPU path) and comes back as numpy.  Boxes with min(w, h) < 1e-3 get
    their row / column zeroed after the kernel, as in the reference (:53-60).
    """
    assert mode in ['iou', 'iof']
    assert type(bboxes1) is type(bboxes2)
    if is_aligned:
        assert bboxes1.shape[0] == bboxes2.shape[0]
    if isinstance(bboxes1, torch.Tensor):
        is_numpy = False
        b1, b2 = bboxes1, bboxes2
    elif isinstance(bboxes1, np.ndarray):
        is_numpy = True
        b1, b2 = _to_device_tensor(bboxes1, device_id), _to_device_tensor(bboxes2, device_id)
    else:
        raise TypeError(f'bboxes must be either a Tensor or numpy array, but got {type(bboxes1)}')

    if b1.numel() == 0 or b2.numel() == 0:
        rows, cols = b1.size(0), b2.size(0)
        out = b1.new_zeros(rows, 1) if is_aligned else b1.new_zeros(rows, cols)
    elif is_aligned:
        out = aligned_obb_overlaps(b1, b2, mode)
    else:
        # the kernel + the reference's `if too_small.any(): outputs[inds] = 0` epilogue in one library call
        # (r3det_obb_overlaps: no host sync, no second pass over the matrix)
        b1 = _as_boxes(b1.contiguous(), "bboxes1")
        b2 = _as_boxes(b2.contiguous(), "bboxes2")
        n1, n2 = b1.size(0), b2.size(0)
        out = b1.new_empty((n1, n2))
        with torch.cuda.device(b1.device):
            ws, wsb = _C.iou_workspace(n1, n2, b1.device)
            _C.check(_C.lib().r3det_obb_overlaps(_C.ptr(b1), n1, _C.ptr(b2), n2, int(mode == 'iou'), _C.ptr(out),
                                                 _C.ptr(ws), wsb, _C.stream()), "obb_overlaps")
    return out.cpu().numpy() if is_numpy else out


def aligned_obb_overlaps(bboxes1, bboxes2, mode='iou'):
    """Pairwise (aligned) overlaps (m, 1), DIFFERENTIABLE w.r.t. the boxes like the reference's
    (box_iou_rotated_wrapper.py:67-92): corners -> edge-edge intersection points + contained
    vertices (24 candidates per pair) -> convex_sort -> shoelace; everything but the index sort
    is torch autograd math.  ``aligned_obb_overlaps_kernel`` is the forward-only value from
    the v3 clipping kernel."""
    area1 = bboxes1[:, 2] * bboxes1[:, 3]
    area2 = bboxes2[:, 2] * bboxes2[:, 3]
    n = bboxes1.size(0)
    p1 = obb2poly(bboxes1).view(n, -1, 2)
    p2 = obb2poly(bboxes2).view(n, -1, 2)
    pts, masks = poly_intersection(p1, p2, area1, area2)
    inter = convex_areas(pts, masks)
    out = inter / (area1 + area2 - inter) if mode == 'iou' else inter / area1
    return out[..., None]


def obb2poly(obboxes):
    """(…, 5) [cx, cy, w, h, theta] -> (…, 8) corners, the reference's sign convention (:95-113):
    half axes (w/2 cos, -w/2 sin) and (-h/2 sin, -h/2 cos)."""
    ctr, w, h, theta = torch.split(obboxes, [2, 1, 1, 1], dim=-1)
    c, s = torch.cos(theta), torch.sin(theta)
    u = torch.cat([w / 2 * c, -w / 2 * s], dim=-1)
    v = torch.cat([-h / 2 * s, -h / 2 * c], dim=-1)
    return torch.cat([ctr + u + v, ctr + u - v, ctr - u - v, ctr - u + v], dim=-1)


def shoelace(pts):
    """Areas of the polygons (…, P, 2) (:116-127)."""
    prev = torch.roll(pts, 1, dims=-2)
    return 0.5 * (pts[..., 0] * prev[..., 1] - prev[..., 0] * pts[..., 1]).sum(dim=-1).abs()


def convex_areas(pts, masks):
    """Area of the convex polygon spanned by the valid points of each row (:130-152): order them with
    convex_sort (closed ring, -1 padding redirected to an appended origin point), shoelace."""
    from .misc import convex_sort
    B, P, _ = pts.shape
    index = convex_sort(pts, masks)
    index = torch.where(index < 0, torch.full_like(index, P), index)
    ring = torch.gather(torch.cat([pts, pts.new_zeros((B, 1, 2))], dim=1), 1, index[..., None].expand(-1, -1, 2))
    cross = ring[:, :-1, 0] * ring[:, 1:, 1] - ring[:, :-1, 1] * ring[:, 1:, 0]
    return 0.5 * cross.sum(dim=-1).abs()


def poly_intersection(pts1, pts2, areas1=None, areas2=None, eps=1e-6):
    """Candidate points of the intersection of two aligned batches of quadrilaterals (B, 4, 2)
    (:155-216): the 16 line-line intersections (valid when both parameters lie in (0, 1)), the
    vertices of each polygon that lie inside the other one (triangle-area test, 1e-3 relative).
    Returns points (B, 24, 2) and the validity mask (B, 24); the masks carry no gradient."""
    e1 = torch.cat([pts1, torch.roll(pts1, -1, dims=1)], dim=2).unsqueeze(2)   # (B, 4, 1, 4)
    e2 = torch.cat([pts2, torch.roll(pts2, -1, dims=1)], dim=2).unsqueeze(1)   # (B, 1, 4, 4)
    x1, y1, x2, y2 = e1.unbind(dim=-1)
    x3, y3, x4, y4 = e2.unbind(dim=-1)
    num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)
    with torch.no_grad():
        den_u = (x2 - x1) * (y1 - y3) - (y2 - y1) * (x1 - x3)
        t, u = den_t / num, den_u / num
        hit = (t > 0) & (t < 1) & (u > 0) & (u < 1)
    t = den_t / (num + eps)
    cross_pts = torch.stack([x1 + t * (x2 - x1), y1 + t * (y2 - y1)], dim=-1)
    B = pts1.size(0)
    with torch.no_grad():
        a1 = shoelace(pts1) if areas1 is None else areas1
        a2 = shoelace(pts2) if areas2 is None else areas2
        tri1 = 0.5 * ((x3 - x1) * (y4 - y1) - (y3 - y1) * (x4 - x1)).abs()    # vertex of 1 with the edges of 2
        in1 = (tri1.sum(dim=-1) - a2[..., None]).abs() < 1e-3 * a2[..., None]
        tri2 = 0.5 * ((x1 - x3) * (y2 - y3) - (x2 - x3) * (y1 - y3)).abs()    # vertex of 2 with the edges of 1
        in2 = (tri2.sum(dim=-2) - a1[..., None]).abs() < 1e-3 * a1[..., None]
    pts = torch.cat([cross_pts.view(B, -1, 2), pts1, pts2], dim=1)
    masks = torch.cat([hit.view(B, -1), in1, in2], dim=1)
    return pts, masks


def aligned_obb_overlaps_kernel(bboxes1, bboxes2, mode='iou'):
    """Forward-only pairwise overlaps (m, 1) through the v3 clipping kernel (exact polygon clipping;
    differs from the differentiable form above by its 1e-3 containment tolerance)."""
    C = _C
    b1 = _as_boxes(bboxes1.contiguous(), "bboxes1")
    b2 = _as_boxes(bboxes2.contiguous(), "bboxes2")
    n = b1.size(0)
    out = b1.new_empty((n,))
    with torch.cuda.device(b1.device):
        C.check(C.lib().r3det_box_iou_rotated_overlaps_aligned(
            C.ptr(b1), C.ptr(b2), n, int(mode == 'iou'), C.ptr(out), C.stream()), "overlaps_aligned")
    return out[:, None]