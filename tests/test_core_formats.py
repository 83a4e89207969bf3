"""CPU: the box coder, the format conversions, the anchor generators and the inside flags of
r3det.core against outputs of the REFERENCE's own functions (tests/golden/heads.npz, written by
tests/golden/make_golden_heads.py from /root/reference).  Elementwise fp32 math on both sides:
compared at 1e-6 relative (torch may fuse differently between releases), integers exactly."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

G = np.load(os.path.join(GOLDEN, "heads.npz"))


def close(a, b, rtol=1e-6, atol=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.all(np.abs(a - b) <= atol + rtol * np.abs(b)), float(np.abs(a - b).max())


def t(name):
    return torch.from_numpy(G[name])


def test_delta2bbox_v1_and_bbox2delta_v1():
    from r3det.core.bbox.coder import DeltaXYWHAOBBoxCoder, bbox2delta_v1, delta2bbox_v1
    rois, deltas, gts = t("coder_rois"), t("coder_deltas"), t("coder_gts")
    assert float(deltas[:, 2:4].abs().max()) > 4.2  # the dw / dh clip at |log(16/1000)| = 4.135 is exercised
    close(delta2bbox_v1(rois, deltas), G["coder_decode"])
    close(delta2bbox_v1(rois, deltas, max_shape=(512, 384)), G["coder_decode_clamped"])
    close(delta2bbox_v1(rois, deltas, (0.1, -0.1, 0.2, 0., 0.05), (0.5, 0.5, 0.25, 0.25, 0.1)), G["coder_decode_stds"])
    close(delta2bbox_v1(rois, t("coder_deltas15")), G["coder_decode15"])
    close(bbox2delta_v1(rois, gts), G["coder_encode"], atol=1e-5)
    close(bbox2delta_v1(rois, gts, (0.1, -0.1, 0.2, 0., 0.05), (0.5, 0.5, 0.25, 0.25, 0.1)), G["coder_encode_stds"],
          atol=1e-5)
    c = DeltaXYWHAOBBoxCoder()
    close(c.decode(rois, c.encode(rois, gts)), G["coder_class_roundtrip"], rtol=1e-5, atol=1e-4)
    with pytest.raises(NotImplementedError):
        DeltaXYWHAOBBoxCoder(angle_range='v2').encode(rois, gts)


@pytest.mark.parametrize("v", ["v1", "v2", "v3"])
def test_rtransforms(v):
    from r3det.core.bbox import rtransforms as rt
    x = t(f"rt_obb_{v}")
    close(rt.obb2hbb(x, v), G[f"rt_obb2hbb_{v}"], atol=1e-4)
    close(rt.obb2poly(x, v), G[f"rt_obb2poly_{v}"], atol=1e-4)
    close(rt.obb2xyxy(x, v), G[f"rt_obb2xyxy_{v}"], atol=1e-4)
    close(rt.poly2obb(t(f"rt_obb2poly_{v}"), v), G[f"rt_poly2obb_{v}"], rtol=1e-5, atol=1e-4)
    x6 = np.hstack([G[f"rt_obb_{v}"], np.linspace(0, 1, 200, dtype=np.float32)[:, None]])
    close(rt.obb2poly_np(x6, v), G[f"rt_obb2poly_np_{v}"], atol=1e-4)
    close(rt.hbb2obb(t(f"rt_obb2xyxy_{v}"), v), G[f"rt_hbb2obb_{v}"], atol=1e-4)
    close(rt.norm_angle(np.linspace(-7, 7, 57), v), G[f"rt_norm_angle_{v}"])
    with pytest.raises(NotImplementedError):
        rt.obb2hbb(x, 'v9')


def test_rbbox2roi_and_rbbox2result():
    from r3det.core import rbbox2result, rbbox2roi
    b = t("rt_obb_v1")
    close(rbbox2roi([b[:7], b[:0], b[7:19]]), G["rt_rbbox2roi"])
    res = rbbox2result(t("rt_result_dets"), t("rt_result_labels"), 15)
    assert [len(r) for r in res] == G["rt_result_sizes"].tolist()
    assert np.array_equal(np.concatenate(res), G["rt_result_cat"])
    empty = rbbox2result(torch.zeros(0, 6), torch.zeros(0, dtype=torch.long), 15)
    assert len(empty) == 15 and all(e.shape == (0, 6) and e.dtype == np.float32 for e in empty)


def test_poly2obb_np_v2_and_cv2_gate():
    """Not in the goldens: the reference's v2 calls np.float (gone in numpy 2) and v1 / v3 need cv2."""
    from r3det.core.bbox import rtransforms as rt
    polys = G["rt_obb2poly_v2"]
    want = G["rt_obb_v2"]
    for p, w in zip(polys[:40], want[:40]):
        got = rt.poly2obb_np(p, 'v2')
        assert got is not None
        close(np.array(got[:4]), w[:4], rtol=1e-4, atol=1e-3)
        d = (got[4] - w[4]) % np.pi  # same axis (a box is symmetric under a half turn), v2 range
        assert min(d, np.pi - d) < 1e-3 or abs(w[2] - w[3]) < 1e-3
        assert -np.pi / 4 <= got[4] < 3 * np.pi / 4
    assert rt.poly2obb_np([0, 0, 1, 0, 1, 1, 0, 1], 'v2') is None  # sides < 2 px
    try:
        import cv2  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            rt.poly2obb_np(polys[0], 'v1')


def test_anchor_generators_and_inside_flags():
    from r3det.core.anchor import PseudoAnchorGenerator, RAnchorGenerator, ranchor_inside_flags
    gen = RAnchorGenerator(strides=[8, 16, 32, 64, 128], ratios=[1.0, 0.5, 2.0], octave_base_scale=4,
                           scales_per_octave=3)
    sizes = [tuple(s) for s in G["anchor_sizes"].tolist()]
    assert gen.num_base_anchors == [9] * 5 and gen.num_levels == 5
    anchors = gen.grid_priors(sizes, device='cpu')
    for i, a in enumerate(anchors):
        assert np.array_equal(a.numpy(), G[f"anchors_l{i}"]), i  # same fp32 operations in the same order
    assert np.array_equal(gen.single_level_grid_priors((128, 128), 0, device='cpu')[:2000].numpy(),
                          G["anchors_1024_l0_head"])
    flags = gen.valid_flags(sizes, (100, 90, 3), device='cpu')
    for i, f in enumerate(flags):
        assert np.array_equal(f.numpy(), G[f"valid_flags_l{i}"])
    fa, vf = torch.cat(anchors), torch.cat(flags)
    assert np.array_equal(ranchor_inside_flags(fa, vf, (100, 90), 0).numpy(), G["inside_b0"])
    assert np.array_equal(ranchor_inside_flags(fa, vf, (100, 90), 16).numpy(), G["inside_b16"])
    assert np.array_equal(ranchor_inside_flags(fa, vf, (100, 90), -1).numpy(), G["inside_bneg"])
    pg = PseudoAnchorGenerator(strides=[8, 16, 32, 64, 128])
    assert pg.num_base_anchors == [1] * 5
    with pytest.raises(NotImplementedError):
        pg.grid_priors(sizes)
    pf = pg.valid_flags(sizes, (100, 90, 3), device='cpu')
    assert [f.numel() for f in pf] == [h * w for h, w in sizes]
    assert all(torch.equal(p, f[::9]) for p, f in zip(pf, flags))
