"""Golden vectors for convex_sort and polygon_iou from the REFERENCE's own CPU code (oracle/_ref,
built from /root/reference by oracle/build.py; run in the build container only):
    python tests/golden/make_golden_rank4.py  ->  tests/golden/rank4.npz
poly_nms has no CPU implementation in the reference: no golden, its oracle is unpinned."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import api as O  # noqa: E402


def polys(n, seed, span=200.0):
    """Rotated rectangles as 8 coordinates in a random cyclic start / orientation, plus a few
    general convex quadrilaterals."""
    r = np.random.default_rng(seed)
    c = r.uniform(0, span, (n, 2))
    w, h = r.uniform(5, 80, n), r.uniform(5, 80, n)
    a = r.uniform(-np.pi, np.pi, n)
    base = np.stack([np.stack([w, h], 1) * s for s in ([.5, .5], [-.5, .5], [-.5, -.5], [.5, -.5])], 1)  # (n,4,2)
    base += r.normal(0, 2.0, base.shape) * (r.random((n, 1, 1)) < 0.3)   # some non-rectangles
    R = np.stack([np.stack([np.cos(a), -np.sin(a)], 1), np.stack([np.sin(a), np.cos(a)], 1)], 1)
    pts = np.einsum('nij,nkj->nki', R, base) + c[:, None]
    for i in range(n):
        k = r.integers(0, 4)
        pts[i] = np.roll(pts[i], k, 0)
        if r.random() < 0.5:
            pts[i] = pts[i][::-1]
    return pts.reshape(n, 8).astype(np.float32)


def main():
    out = {}
    a, b = polys(60, 1), polys(50, 2)
    b[:5] = a[:5]                       # identical polygons
    b[5] = a[5] + 1e-3                  # near-identical
    out["poly_a"], out["poly_b"] = a, b
    out["poly_iou"] = O.ref_polygon_iou(a, b)
    r = np.random.default_rng(3)
    for P in (4, 8, 24):
        pts = r.uniform(0, 100, (100, P, 2)).astype(np.float32)
        masks = r.random((100, P)) < 0.8
        masks[:, 0] |= ~masks.any(1)    # at least one valid point
        pts[::7, 1] = pts[::7, 0]       # duplicated points
        pts[::11, :, 1] = 5.0           # all collinear
        for circ in (True, False):
            out[f"cs_pts_{P}"] = pts
            out[f"cs_masks_{P}"] = masks
            out[f"cs_idx_{P}_{int(circ)}"] = O.ref_convex_sort(pts, masks, circ)
    np.savez_compressed(os.path.join(HERE, "rank4.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
