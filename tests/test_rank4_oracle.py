"""CPU: the oracle's restatements of polygon_iou / convex_sort against vectors recorded from the
REFERENCE's own CPU code (tests/golden/rank4.npz, make_golden_rank4.py), and -- when oracle/_ref is
present -- against that code directly on fresh inputs.  poly_nms has no CPU reference (unpinned):
only self-consistency properties are checked for its IoU."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from oracle import api as O


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "rank4.npz"))


def sort_keys(pts, masks):
    """fp32 keys of convex_sort's prologue (to tell rows with tied keys apart)."""
    m = masks.astype(np.float32)
    my = m * pts[..., 1] + (1 - m) * np.float32(1e7)
    start = my.argmin(1)
    s = pts[np.arange(len(pts)), start]
    d = pts - s[:, None]
    key = d[..., 0] / np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + np.float32(1e-6))
    return key.astype(np.float32)


def tie_free(pts, masks):
    key = sort_keys(pts, masks)
    ks = np.sort(key, 1)
    return (np.diff(ks, axis=1) != 0).all(1)


def test_polygon_iou_matches_reference_golden(g):
    got = O.polygon_iou(g["poly_a"], g["poly_b"])
    assert np.array_equal(got, g["poly_iou"])
    assert (got > 0).sum() > 100 and got.max() <= 1.0
    assert np.allclose(np.diag(got[:5, :5]), 1.0, atol=1e-5)  # identical polygons


@pytest.mark.parametrize("P", [4, 8, 24])
@pytest.mark.parametrize("circular", [True, False])
def test_convex_sort_matches_reference_golden(g, P, circular):
    """Rows whose sort keys are all distinct must match exactly; the order of EQUAL keys is left to
    torch.argsort in the reference (unstable) and is index order here."""
    pts, masks = g[f"cs_pts_{P}"], g[f"cs_masks_{P}"]
    got = O.convex_sort(pts, masks, circular)
    want = g[f"cs_idx_{P}_{int(circular)}"]
    ok = tie_free(pts, masks)
    assert ok.sum() >= 60
    assert np.array_equal(got[ok], want[ok])


@pytest.mark.skipif(not O.ref_rank4_available(), reason="oracle/_ref not present")
def test_against_reference_build_fresh_inputs():
    r = np.random.default_rng(11)
    a = r.uniform(0, 60, (40, 4, 2)).astype(np.float32)
    a += r.uniform(0, 100, (40, 1, 2)).astype(np.float32)
    assert np.array_equal(O.polygon_iou(a.reshape(40, 8), a[::-1].reshape(40, 8)),
                          O.ref_polygon_iou(a.reshape(40, 8), a[::-1].reshape(40, 8)))
    pts = r.uniform(0, 50, (300, 12, 2)).astype(np.float32)
    masks = r.random((300, 12)) < 0.7
    masks[:, 0] = True
    ok = tie_free(pts, masks)
    for circ in (True, False):
        assert np.array_equal(O.convex_sort(pts, masks, circ)[ok], O.ref_convex_sort(pts, masks, circ)[ok])


def test_convex_sort_properties():
    r = np.random.default_rng(2)
    pts = r.uniform(0, 100, (200, 16, 2)).astype(np.float32)
    masks = np.ones((200, 16), bool)
    idx = O.convex_sort(pts, masks, True)
    for b in range(200):
        row = idx[b]
        end = 1 + int(np.nonzero(row[1:] == row[0])[0][0])            # closing entry; what follows is stale
        h = row[:end + 1]                                             # (popped slots are not reset, as in
        assert (h >= 0).all() and len(set(h[:-1])) == len(h) - 1      #  the reference: convex_cpu.cpp:66-83)
        assert pts[b, h[0], 1] == pts[b, :, 1].min()                  # starts at the lowest point
        poly = pts[b, h[:-1]]
        x, y = poly[:, 0].astype(np.float64), poly[:, 1].astype(np.float64)
        cr = (np.roll(x, -1) - x) * (np.roll(y, -2) - np.roll(y, -1)) - (np.roll(y, -1) - y) * (np.roll(x, -2) - np.roll(x, -1))
        assert (cr >= -1e-3).all()                                    # counter-clockwise convex chain
    # masked points never appear
    masks[:, 5] = False
    assert not (O.convex_sort(pts, masks, False) == 5).any()


def test_poly_nms_iou_properties():
    """devPolyIoU restatement (unpinned): symmetric, 1 on identical convex quads, close to the
    polygon_iou value on generic rectangles, ~0 when disjoint."""
    r = np.random.default_rng(5)
    c = r.uniform(0, 100, (30, 2))
    w, h, a = r.uniform(10, 40, 30), r.uniform(10, 40, 30), r.uniform(0, np.pi, 30)
    base = np.stack([np.stack([w, h], 1) * s for s in ([.5, .5], [-.5, .5], [-.5, -.5], [.5, -.5])], 1)
    R = np.stack([np.stack([np.cos(a), -np.sin(a)], 1), np.stack([np.sin(a), np.cos(a)], 1)], 1)
    polys = (np.einsum('nij,nkj->nki', R, base) + c[:, None]).reshape(30, 8).astype(np.float32)
    m = O.poly_iou_mat(polys, polys)
    assert np.allclose(m, m.T, atol=1e-5)
    assert np.allclose(np.diag(m), 1.0, atol=1e-5)
    assert np.abs(m - O.polygon_iou(polys, polys)).max() < 2e-3   # two different algorithms
    far = polys.copy()
    far[:, 0::2] += 1000
    assert np.abs(O.poly_iou_mat(polys, far)).max() < 1e-3  # signed fan areas cancel up to fp32 rounding
    dets = np.hstack([polys, r.uniform(0, 1, (30, 1)).astype(np.float32)])
    keep = O.poly_nms(dets, 0.1)
    assert len(keep) >= 1 and len(set(keep)) == len(keep)
    assert (np.diff(dets[keep, 8]) <= 0).all()
