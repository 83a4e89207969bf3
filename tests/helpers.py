"""Shared input generators for the tests (seeded, fp32)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rand_boxes(n, seed, span=1024.0, lo=8.0, hi=128.0, amin=-np.pi / 2, amax=0.0):
    r = np.random.default_rng(seed)
    return np.stack([r.uniform(0, span, n), r.uniform(0, span, n), r.uniform(lo, hi, n),
                     r.uniform(lo, hi, n), r.uniform(amin, amax, n)], 1).astype(np.float32)


def anchor_grid(size=1024, strides=(8, 16, 32, 64, 128)):
    """The 196 416 RetinaNet anchors of a size x size input as (cx,cy,w,h,0) rows
    (octave_base_scale 4, 3 scales, ratios [1, .5, 2]; position-major, anchor-minor)."""
    out = []
    for s in strides:
        scales = np.array([4 * 2 ** (i / 3) for i in range(3)])
        ratios = np.array([1.0, 0.5, 2.0])
        h_r = np.sqrt(ratios)
        w_r = 1 / h_r
        ws = (s * w_r[:, None] * scales[None, :]).reshape(-1)
        hs = (s * h_r[:, None] * scales[None, :]).reshape(-1)
        f = size // s
        xs, ys = np.meshgrid(np.arange(f) * s, np.arange(f) * s)
        ctr = np.stack([xs.reshape(-1), ys.reshape(-1)], 1).astype(np.float64)
        a = np.zeros((f * f, 9, 5))
        a[:, :, 0:2] = ctr[:, None, :]
        a[:, :, 2] = ws[None]
        a[:, :, 3] = hs[None]
        out.append(a.reshape(-1, 5))
    return np.concatenate(out).astype(np.float32)


def dota_like_gt(n, seed, size=1024):
    r = np.random.default_rng(seed)
    w = np.exp(r.uniform(np.log(10), np.log(300), n))
    asp = np.exp(r.uniform(0, np.log(8), n))
    h = np.maximum(w / asp, 4)
    return np.stack([r.uniform(0, size, n), r.uniform(0, size, n), w, h,
                     r.uniform(-np.pi / 2, 0, n)], 1).astype(np.float32)


def fr_boxes(N, H, W, stride, seed, jitter=0.1, adversarial=False):
    """Per-position boxes (N*H*W, 5): centres near their cell (realistic) or uniform random."""
    r = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    cx = (xs.reshape(-1) * stride).astype(np.float64)
    cy = (ys.reshape(-1) * stride).astype(np.float64)
    n = N * H * W
    b = np.zeros((n, 5))
    if adversarial:
        b[:, 0] = r.uniform(-2 * stride, (W + 2) * stride, n)
        b[:, 1] = r.uniform(-2 * stride, (H + 2) * stride, n)
    else:
        b[:, 0] = np.tile(cx, N) + r.normal(0, jitter * 4 * stride, n)
        b[:, 1] = np.tile(cy, N) + r.normal(0, jitter * 4 * stride, n)
    b[:, 2] = r.uniform(2 * stride, 8 * stride, n)
    b[:, 3] = r.uniform(2 * stride, 8 * stride, n)
    b[:, 4] = r.uniform(-np.pi / 2, 0, n)
    return b.astype(np.float32)


def assert_pool_matches_golden(got_boxes, got_scores, want_boxes, want_scores, level_rows, tag=""):
    """Compare one image's pre-NMS pool (n, 5) / (n, C + 1) with the arrays recorded from the reference's
    ``_get_bboxes_single(with_nms=False)`` (tests/golden/getbboxes.npz).  Bars: scores <= 1e-6 absolute, boxes
    <= 1e-5 relative; inside every level slice (``level_rows`` rows each) row set AND order exact where the
    recorded best scores are distinct -- rows of a run of exactly equal best scores (topk leaves their order
    open) are compared as a multiset."""
    gb, gs = np.asarray(got_boxes, np.float32), np.asarray(got_scores, np.float32)
    wb, ws = np.asarray(want_boxes, np.float32), np.asarray(want_scores, np.float32)
    assert gb.shape == wb.shape and gs.shape == ws.shape, (tag, gb.shape, wb.shape, gs.shape, ws.shape)
    assert sum(level_rows) == wb.shape[0], (tag, level_rows, wb.shape)
    assert np.all(gs[:, -1] == 0) and np.all(ws[:, -1] == 0), tag  # the background column
    off = 0
    for rows in level_rows:
        sl = slice(off, off + rows)
        off += rows
        g = np.concatenate([gs[sl], gb[sl]], 1)
        w = np.concatenate([ws[sl], wb[sl]], 1)
        best = ws[sl, :-1].max(1)
        # runs of equal best score (adjacent after the reference's topk; a level without a top-k is in row order
        # and is compared row by row)
        i = 0
        while i < rows:
            j = i + 1
            while j < rows and best[j] == best[i]:
                j += 1
            if j - i > 1:
                # canonical order inside a tie run: lexicographic on the recorded row contents
                g[i:j] = g[i:j][np.lexsort(np.round(g[i:j, ::-1].T.astype(np.float64), 3))]
                w[i:j] = w[i:j][np.lexsort(np.round(w[i:j, ::-1].T.astype(np.float64), 3))]
            i = j
        nc = gs.shape[1]
        ds = np.abs(g[:, :nc] - w[:, :nc]).max()
        assert ds <= 1e-6, (tag, "scores", float(ds))
        assert np.allclose(g[:, nc:], w[:, nc:], rtol=1e-5, atol=1e-5), (tag, "boxes", float(np.abs(g[:, nc:] - w[:, nc:]).max()))
