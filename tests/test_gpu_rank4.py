"""GPU parity for the ops outside the shipped configs: polygon_iou, convex_sort
(exact vs the oracle, which the reference's own CPU code pins -- tests/test_rank4_oracle.py), and
poly_nms (exact keep vs the oracle's restatement of poly_nms_cuda.cu; IoU matrix <= 1e-5)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from oracle import api as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def quads(n, seed, span=300.0):
    r = np.random.default_rng(seed)
    c = r.uniform(0, span, (n, 2))
    w, h, a = r.uniform(8, 90, n), r.uniform(8, 90, n), r.uniform(-np.pi, np.pi, n)
    base = np.stack([np.stack([w, h], 1) * s for s in ([.5, .5], [-.5, .5], [-.5, -.5], [.5, -.5])], 1)
    R = np.stack([np.stack([np.cos(a), -np.sin(a)], 1), np.stack([np.sin(a), np.cos(a)], 1)], 1)
    return (np.einsum('nij,nkj->nki', R, base) + c[:, None]).reshape(n, 8).astype(np.float32)


def test_polygon_iou_golden_and_oracle():
    from r3det.ops import polygon_iou
    g = np.load(os.path.join(GOLDEN, "rank4.npz"))
    got = polygon_iou(dev(g["poly_a"]), dev(g["poly_b"]))
    assert np.array_equal(got.cpu().numpy(), g["poly_iou"])           # the reference CPU code's output
    a, b = quads(300, 1), quads(257, 2)
    assert np.array_equal(polygon_iou(dev(a), dev(b)).cpu().numpy(), O.polygon_iou(a, b))
    # reference calling convention: CPU tensors in, CPU tensor out (polygon_geo_cpu CHECK_CPU)
    out = polygon_iou(torch.from_numpy(a[:7]), torch.from_numpy(b[:9]))
    assert out.device.type == 'cpu' and np.array_equal(out.numpy(), O.polygon_iou(a[:7], b[:9]))
    assert polygon_iou(dev(a[:0]), dev(b)).shape == (0, 257)


@pytest.mark.parametrize("P", [4, 8, 24])
@pytest.mark.parametrize("circular", [True, False])
def test_convex_sort(P, circular):
    from r3det.ops import convex_sort
    g = np.load(os.path.join(GOLDEN, "rank4.npz"))
    pts, masks = g[f"cs_pts_{P}"], g[f"cs_masks_{P}"]
    got = convex_sort(dev(pts), dev(masks), circular).cpu().numpy()
    assert np.array_equal(got, O.convex_sort(pts, masks, circular))   # incl. the tied-key rows
    r = np.random.default_rng(P)
    pts = r.uniform(0, 64, (5000, P, 2)).astype(np.float32)
    masks = r.random((5000, P)) < 0.75
    masks[:, 0] = True
    assert np.array_equal(convex_sort(dev(pts), dev(masks), circular).cpu().numpy(),
                          O.convex_sort(pts, masks, circular))
    assert convex_sort(dev(pts[:0]), dev(masks[:0]), circular).shape == (0, P + int(circular))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 700, 3000])
def test_poly_nms(n):
    from r3det.ops import poly_nms
    polys = quads(n, 40 + n, span=400.0 if n > 100 else 120.0)
    s = np.random.default_rng(n).uniform(0, 1, n).astype(np.float32)
    dets = np.hstack([polys, s[:, None]])
    for thr in (0.1, 0.5):
        want = O.poly_nms(dets, thr)
        d, keep = poly_nms(dev(dets), thr)
        assert np.array_equal(keep.cpu().numpy(), want)
        assert np.array_equal(d.cpu().numpy(), dets[want])
    out, keep = poly_nms(dets, 0.3, device_id=0)                      # numpy in -> numpy out
    assert isinstance(keep, np.ndarray) and np.array_equal(keep, O.poly_nms(dets, 0.3))
    with pytest.raises(NotImplementedError):
        poly_nms(torch.from_numpy(dets), 0.3)                         # CPU tensors: as in the reference


def test_poly_iou_matrix_vs_oracle():
    from r3det import _C
    a, b = quads(120, 7, span=150.), quads(90, 8, span=150.)
    out = torch.empty(120, 90, device='cuda')
    ta, tb = dev(a), dev(b)
    _C.check(_C.lib().r3det_poly_iou_mat(_C.ptr(ta), 120, 8, _C.ptr(tb), 90, 8, _C.ptr(out), _C.stream()),
             "poly_iou_mat")
    want = O.poly_iou_mat(a, b)
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-5


def test_aligned_obb_overlaps_differentiable():
    """obb_overlaps(is_aligned=True): the reference's differentiable torch formulation
    (box_iou_rotated_wrapper.py:67-216) on top of convex_sort.  Value: close to the exact v3 clipping
    kernel wherever the pair really overlaps (the formulation has a 1e-3 containment tolerance);
    gradient: matches central differences of the function itself."""
    from r3det.ops.iou import aligned_obb_overlaps, aligned_obb_overlaps_kernel, obb_overlaps
    r = np.random.default_rng(9)
    n = 400
    b1 = np.stack([r.uniform(40, 60, n), r.uniform(40, 60, n), r.uniform(15, 40, n), r.uniform(15, 40, n),
                   r.uniform(-1.5, 1.5, n)], 1).astype(np.float32)
    b2 = b1 + np.stack([r.normal(0, 6, n), r.normal(0, 6, n), r.normal(0, 4, n), r.normal(0, 4, n),
                        r.normal(0, 0.3, n)], 1).astype(np.float32)
    t1, t2 = dev(b1), dev(b2)
    for mode in ('iou', 'iof'):
        v = obb_overlaps(t1, t2, mode=mode, is_aligned=True)
        assert v.shape == (n, 1)
        k = aligned_obb_overlaps_kernel(t1, t2, mode)
        assert (v - k).abs().max().item() < 5e-3
    # gradient of the IoU w.r.t. the first box, against central differences in fp64-ish steps
    t1g = t1[:50].clone().requires_grad_(True)
    out = aligned_obb_overlaps(t1g, t2[:50]).sum()
    out.backward()
    g = t1g.grad.cpu().numpy()
    assert np.isfinite(g).all() and np.abs(g).sum() > 0
    eps = 1e-2
    for col in (0, 2, 4):
        d = torch.zeros_like(t1[:50])
        d[:, col] = eps
        fd = (aligned_obb_overlaps(t1[:50] + d, t2[:50]) - aligned_obb_overlaps(t1[:50] - d, t2[:50]))[:, 0] / (2 * eps)
        ok = np.abs(fd.cpu().numpy() - g[:, col]) < 5e-2 * (1 + np.abs(g[:, col]))
        assert ok.mean() > 0.9   # (topology changes inside the finite-difference step break a few)
