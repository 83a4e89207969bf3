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
