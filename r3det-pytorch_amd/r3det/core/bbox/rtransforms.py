"""Box-format conversions either side of the rotated ops (core/bbox/rtransforms.py).

Formats: obb ``(cx, cy, w, h, angle rad)`` in the three angle conventions of the reference
('v1': angle in [-pi/2, 0); 'v2': [-pi/4, 3pi/4); 'v3': [-pi/2, pi/2)), polygons
``(x0, y0, ..., x3, y3)``, horizontal boxes ``(x1, y1, x2, y2)``.  Every function dispatches on
``version`` like the reference's (:50-182) and is checked against the reference functions
themselves through tests/golden/heads.npz (tests/test_core_formats.py).  On the hot path:
``obb2hbb(gt, 'v1')`` feeds the stage-0 assignment (rotate_anchor_head.py:220-224) and
``rbbox2result`` closes ``simple_test`` (r3det.py:137-141).
"""
import math

import numpy as np
import torch

_HALF_PI = np.pi / 2


def _dispatch(table, version, *args):
    if version not in table:
        raise NotImplementedError
    return table[version](*args)


def rbbox2result(bboxes, labels, num_classes):
    """(n, 6) dets + (n,) labels -> list over classes of float32 ndarrays (:10-24)."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 6), dtype=np.float32) for _ in range(num_classes)]
    bboxes, labels = bboxes.cpu().numpy(), labels.cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


def rbbox2roi(bbox_list):
    """list over images of (n_i, >=5) -> (sum n_i, 6) rows ``(image index, cx, cy, w, h, a)`` (:27-45)."""
    rois = [torch.cat([b.new_full((b.size(0), 1), i), b[:, :5]], dim=-1) if b.size(0) > 0 else b.new_zeros((0, 6))
            for i, b in enumerate(bbox_list)]
    return torch.cat(rois, 0)


def norm_angle(angle, angle_range):
    """Wrap into the convention's range (:606-622); 'v1' is left alone."""
    if angle_range == 'v1':
        return angle
    if angle_range == 'v2':
        return (angle + np.pi / 4) % np.pi - np.pi / 4
    if angle_range == 'v3':
        return (angle + _HALF_PI) % np.pi - _HALF_PI
    print('Not yet implemented.')


# ---------------------------------------------------------------------------------- obb -> polygon
def _half_axes(w, h, a, cos, sin):
    """Half-width vector along the box's w axis and half-height vector along its h axis."""
    c, s = cos(a), sin(a)
    return w / 2 * c, w / 2 * s, -h / 2 * s, h / 2 * c


def _corners_v1(x, y, w, h, a, cos, sin):
    wx, wy, hx, hy = _half_axes(w, h, a, cos, sin)
    return [x - wx - hx, y - wy - hy, x + wx - hx, y + wy - hy, x + wx + hx, y + wy + hy, x - wx + hx, y - wy + hy]


def obb2poly_v1(rboxes):
    return torch.stack(_corners_v1(*rboxes[:, :5].unbind(1), torch.cos, torch.sin), dim=-1)  # (:344-365)


def obb2poly_v2(rboxes):
    """Rotation matrix times the axis-aligned corner set tl, tr, br, bl (:368-415, v3 is the same code)."""
    x, y, w, h, a = rboxes[:, :5].unbind(1)
    lx, ly, rx, ry = -w * 0.5, -h * 0.5, w * 0.5, h * 0.5
    rects = torch.stack([torch.stack([lx, rx, rx, lx], 1), torch.stack([ly, ly, ry, ry], 1)], 1)  # (N, 2, 4)
    s, c = torch.sin(a), torch.cos(a)
    M = torch.stack([torch.stack([c, -s], 1), torch.stack([s, c], 1)], 1)                              # (N, 2, 2)
    p = M.matmul(rects)                                                                                # (N, 2, 4)
    polys = p.permute(0, 2, 1).reshape(-1, 8).clone()
    polys[:, 0::2] += x[:, None]
    polys[:, 1::2] += y[:, None]
    return polys.contiguous()


obb2poly_v3 = obb2poly_v2


def obb2poly(rbboxes, version='v1'):
    return _dispatch({'v1': obb2poly_v1, 'v2': obb2poly_v2, 'v3': obb2poly_v3}, version, rbboxes)


def obb2poly_np_v1(rbboxes):
    """(n, 6) ``[obb5, score]`` -> (n, 9) ``[poly8, score]`` (:553-575)."""
    return np.stack(_corners_v1(*(rbboxes[:, i] for i in range(5)), np.cos, np.sin) + [rbboxes[:, 5]], axis=-1)


def _best_begin_point(row):
    """Rotate the vertex order so that it is closest (sum of distances) to tl, tr, br, bl of the
    polygon's bounding box; first minimum wins (:622-660)."""
    pts, score = [list(row[2 * i:2 * i + 2]) for i in range(4)], row[8]
    xs, ys = [p[0] for p in pts], [p[1] for p in pts]
    target = [[min(xs), min(ys)], [max(xs), min(ys)], [max(xs), max(ys)], [min(xs), max(ys)]]
    best, best_cost = 0, 100000000.0
    for r in range(4):
        cost = sum(math.sqrt(math.pow(pts[(r + k) % 4][0] - target[k][0], 2)
                             + math.pow(pts[(r + k) % 4][1] - target[k][1], 2)) for k in range(4))
        if cost < best_cost:
            best, best_cost = r, cost
    return np.hstack((np.array([pts[(best + k) % 4] for k in range(4)]).reshape(8), np.array(score)))


def obb2poly_np_v2(rrects):
    """(:578-603): per box rotation in float64, rows cast to float32, then the begin-point rule."""
    polys = []
    for x, y, w, h, a, score in (r[:6] for r in rrects):
        rect = np.array([[-w / 2, w / 2, w / 2, -w / 2], [-h / 2, -h / 2, h / 2, h / 2]])
        R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        p = R.dot(rect)
        polys.append(np.array([p[0, 0] + x, p[1, 0] + y, p[0, 1] + x, p[1, 1] + y, p[0, 2] + x, p[1, 2] + y,
                               p[0, 3] + x, p[1, 3] + y, score], dtype=np.float32))
    polys = np.array(polys)
    return np.array([_best_begin_point(r) for r in polys.tolist()])


def obb2poly_np_v3(obboxes):
    """(:606-627) incl. its catch-all: input that cannot be split gives one zero row."""
    try:
        center, w, h, theta, score = np.split(obboxes, (2, 3, 4, 5), axis=-1)
    except Exception:  # noqa: BLE001
        return np.zeros((1, 9))
    c, s = np.cos(theta), np.sin(theta)
    v1 = np.concatenate([w / 2 * c, -w / 2 * s], axis=-1)
    v2 = np.concatenate([-h / 2 * s, -h / 2 * c], axis=-1)
    return np.concatenate([center + v1 + v2, center + v1 - v2, center - v1 - v2, center - v1 + v2, score], axis=-1)


def obb2poly_np(rbboxes, version='v1'):
    return _dispatch({'v1': obb2poly_np_v1, 'v2': obb2poly_np_v2, 'v3': obb2poly_np_v3}, version, rbboxes)


# ---------------------------------------------------------------------------------- obb -> horizontal
def _extent_v1(rbboxes):
    """Width / height of the circumscribed horizontal box for angles in [-pi/2, 0] (cos >= 0 >= sin)."""
    w, h, a = rbboxes[:, 2::5], rbboxes[:, 3::5], rbboxes[:, 4::5]
    c, s = torch.cos(a), torch.sin(a)
    return c * w - s * h, -s * w + c * h


def obb2hbb_v1(rbboxes):
    """Same centre, ``(w, h) <- (extent_h, extent_w)``, angle -pi/2 (:443-462)."""
    ew, eh = _extent_v1(rbboxes)
    out = rbboxes.clone().detach()
    out[:, 2::5] = eh
    out[:, 3::5] = ew
    out[:, 4::5] = -_HALF_PI
    return out


def obb2xyxy_v2(rboxes):
    polys = obb2poly_v2(rboxes)
    xs, ys = polys[:, 0::2], polys[:, 1::2]
    return torch.stack([xs.min(1)[0], ys.min(1)[0], xs.max(1)[0], ys.max(1)[0]], dim=1)  # (:530-543)


def obb2hbb_v2(rboxes):
    """Bounding box of the polygon as an obb whose long side is w: angle 0, or pi/2 for tall boxes (:465-489)."""
    b = obb2xyxy_v2(rboxes)
    cx, cy = (b[:, 2] + b[:, 0]) / 2.0, (b[:, 3] + b[:, 1]) / 2.0
    e1, e2 = (b[:, 2] - b[:, 0]).abs(), (b[:, 3] - b[:, 1]).abs()
    tall = e1 < e2
    out = torch.stack((cx, cy, e1, e2, b.new_zeros(b.size(0))), dim=1)
    out[tall, 2] = e2[tall]
    out[tall, 3] = e1[tall]
    out[tall, 4] = np.pi / 2.0
    return out


def obb2xyxy_v3(obboxes):
    center, w, h, theta = torch.split(obboxes, [2, 1, 1, 1], dim=-1)
    c, s = torch.cos(theta), torch.sin(theta)
    bias = torch.cat([(w / 2 * c).abs() + (h / 2 * s).abs(), (w / 2 * s).abs() + (h / 2 * c).abs()], dim=-1)
    return torch.cat([center - bias, center + bias], dim=-1)  # (:546-560)


def _xyxy2obb(hb, tall_angle):
    x, y = (hb[..., 0] + hb[..., 2]) * 0.5, (hb[..., 1] + hb[..., 3]) * 0.5
    w, h = hb[..., 2] - hb[..., 0], hb[..., 3] - hb[..., 1]
    t = x.new_zeros(*x.shape)
    return torch.where((w >= h)[..., None], torch.stack([x, y, w, h, t], dim=-1),
                       torch.stack([x, y, h, w, t + tall_angle], dim=-1))


def obb2hbb_v3(obboxes):
    return _xyxy2obb(obb2xyxy_v3(obboxes), -_HALF_PI)  # (:515-535)


def obb2hbb(rbboxes, version='v1'):
    return _dispatch({'v1': obb2hbb_v1, 'v2': obb2hbb_v2, 'v3': obb2hbb_v3}, version, rbboxes)


def obb2xyxy_v1(rbboxes):
    ew, eh = _extent_v1(rbboxes)
    dx, dy, dw, dh = rbboxes[..., 0], rbboxes[..., 1], ew.reshape(-1), eh.reshape(-1)
    return torch.stack((dx - dw / 2, dy - dh / 2, dx + dw / 2, dy + dh / 2), -1)  # (:502-527)


def obb2xyxy(rbboxes, version='v1'):
    return _dispatch({'v1': obb2xyxy_v1, 'v2': obb2xyxy_v2, 'v3': obb2xyxy_v3}, version, rbboxes)


# ---------------------------------------------------------------------------------- horizontal -> obb
def hbb2obb_v1(hbboxes):
    """(n, 4k) -> (n, k, 5) rows ``(cx, cy, h, w, -pi/2)`` (:538-552)."""
    x, y = (hbboxes[:, 0::4] + hbboxes[:, 2::4]) * 0.5, (hbboxes[:, 1::4] + hbboxes[:, 3::4]) * 0.5
    w, h = hbboxes[:, 2::4] - hbboxes[:, 0::4], hbboxes[:, 3::4] - hbboxes[:, 1::4]
    return torch.stack([x, y, h, w, x.new_zeros(*x.shape) - _HALF_PI], dim=-1)


def hbb2obb_v2(hbboxes):
    return _xyxy2obb(hbboxes, _HALF_PI)  # (:555-571)


def hbb2obb_v3(hbboxes):
    return _xyxy2obb(hbboxes, -_HALF_PI)  # (:574-590)


def hbb2obb(hbboxes, version='v1'):
    return _dispatch({'v1': hbb2obb_v1, 'v2': hbb2obb_v2, 'v3': hbb2obb_v3}, version, hbboxes)


# ---------------------------------------------------------------------------------- polygon -> obb
def poly2obb_v1(polys):
    """Edge p0p1 is w, p1p2 is h, angle from p0p1 folded into [-pi/2, 0) with w/h swapped on odd
    quarter turns (:185-208)."""
    p = polys.reshape(-1, 4, 2)
    cx, cy = p[:, :, 0].sum(1, keepdim=True) / 4., p[:, :, 1].sum(1, keepdim=True) / 4.
    e1 = torch.norm(p[:, 0] - p[:, 1], dim=-1).unsqueeze(1)
    e2 = torch.norm(p[:, 1] - p[:, 2], dim=-1).unsqueeze(1)
    t = torch.atan2(-(p[:, 1, 0] - p[:, 0, 0]), p[:, 1, 1] - p[:, 0, 1]).unsqueeze(1)
    even = torch.eq(torch.remainder((t / (-np.pi * 0.5)).floor_(), 2), 0)
    return torch.cat([cx, cy, torch.where(even, e2, e1), torch.where(even, e1, e2),
                      torch.remainder(t, -np.pi * 0.5)], dim=1)


def _poly2obb_long_edge(polys, angle_range):
    """w = longer of the first two edges, angle = direction of that edge, wrapped (:211-277)."""
    polys = polys.reshape(-1, 8)
    p1, p2, p3, p4 = polys[..., :8].chunk(4, 1)
    e1 = torch.sqrt(torch.pow(p1[..., 0] - p2[..., 0], 2) + torch.pow(p1[..., 1] - p2[..., 1], 2))
    e2 = torch.sqrt(torch.pow(p2[..., 0] - p3[..., 0], 2) + torch.pow(p2[..., 1] - p3[..., 1], 2))
    a1 = torch.atan2(p2[..., 1] - p1[..., 1], p2[..., 0] - p1[..., 0])
    a2 = torch.atan2(p4[..., 1] - p1[..., 1], p4[..., 0] - p1[..., 0])
    ang = norm_angle(torch.where(e1 > e2, a1, a2), angle_range)
    edges = torch.stack([e1, e2], dim=1)
    return torch.stack([(p1[..., 0] + p3[..., 0]) / 2.0, (p1[..., 1] + p3[..., 1]) / 2.0, edges.max(1)[0],
                        edges.min(1)[0], ang], 1)


def poly2obb_v2(polys):
    return _poly2obb_long_edge(polys, 'v2')


def poly2obb_v3(polys):
    return _poly2obb_long_edge(polys, 'v3')


def poly2obb(polys, version='v1'):
    return _dispatch({'v1': poly2obb_v1, 'v2': poly2obb_v2, 'v3': poly2obb_v3}, version, polys)


def _min_area_rect(pts):
    try:
        import cv2
    except ImportError as e:  # dataset-side helper: needs OpenCV exactly as the reference does (:3,288,329)
        raise ImportError('poly2obb_np (v1 / v3) needs cv2.minAreaRect') from e
    (x, y), (w, h), a = cv2.minAreaRect(pts)
    return x, y, w, h, a


def poly2obb_np_v1(poly):
    """Minimum-area rectangle, angle (degrees) folded into [-90, 0); None for sides < 2 px (:280-303)."""
    x, y, w, h, a = _min_area_rect(np.array(poly).reshape((4, 2)))
    if w < 2 or h < 2:
        return
    while not 0 > a >= -90:
        a, w, h = (a - 90, h, w) if a >= 0 else (a + 90, h, w)
    a = a / 180 * np.pi
    assert 0 > a >= -np.pi / 2
    return x, y, w, h, a


def poly2obb_np_v2(poly):
    """Long-edge rule in numpy (:306-336)."""
    poly = np.array(poly[:8], dtype=np.float32)
    p1, p2, p3, p4 = ((poly[2 * i], poly[2 * i + 1]) for i in range(4))
    e1 = np.sqrt((p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]))
    e2 = np.sqrt((p2[0] - p3[0]) * (p2[0] - p3[0]) + (p2[1] - p3[1]) * (p2[1] - p3[1]))
    if e1 < 2 or e2 < 2:
        return
    ref = p2 if e1 > e2 else p4
    angle = norm_angle(np.arctan2(float(ref[1] - p1[1]), float(ref[0] - p1[0])), 'v2')
    return float(p1[0] + p3[0]) / 2, float(p1[1] + p3[1]) / 2, max(e1, e2), min(e1, e2), angle


def poly2obb_np_v3(poly):
    """Minimum-area rectangle with w the long side, angle in [-pi/2, pi/2) (:339-361)."""
    x, y, w, h, a = _min_area_rect(np.array(poly).reshape((4, 2)))
    if w < 2 or h < 2:
        return
    a = -a / 180 * np.pi
    if w < h:
        w, h, a = h, w, a + np.pi / 2
    while not np.pi / 2 > a >= -np.pi / 2:
        a = a - np.pi if a >= np.pi / 2 else a + np.pi
    return x, y, w, h, a


def poly2obb_np(polys, version='v1'):
    return _dispatch({'v1': poly2obb_np_v1, 'v2': poly2obb_np_v2, 'v3': poly2obb_np_v3}, version, polys)
