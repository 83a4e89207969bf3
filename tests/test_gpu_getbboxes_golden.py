"""GPU: r3det_level_pool (through RRetinaHead.decode_bboxes) and the whole get_bboxes of both heads against
arrays recorded from the REFERENCE's own RAnchorHead.get_bboxes / _get_bboxes_single and
RRetinaRefineHead.get_bboxes (tests/golden/getbboxes.npz; rotate_anchor_head.py:499-680,
rotate_retina_refine_head.py:147-200).  NCHW and channels_last maps, shared anchors (first stage) and per-image
rois (refine stage), levels with and without a top-k cut.  Bars: scores <= 1e-6, boxes <= 1e-5 relative, row set and
order exact where the recorded scores are distinct (helpers.assert_pool_matches_golden)."""
import numpy as np
import pytest
import torch

from helpers import assert_pool_matches_golden
from test_getbboxes_golden import G, IMG, heads, level_rows, maps, rois

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("stage,nms_pre", [("s0", 500), ("s0", -1), ("sr", 100), ("sr", -1)])
def test_level_pool_matches_reference_pool(stage, nms_pre, channels_last):
    h0, hr = heads()
    head, A = (h0.cuda(), 9) if stage == "s0" else (hr.cuda(), 1)
    cls, reg = maps(stage, "cuda", channels_last)
    cfg = dict(nms_pre=nms_pre, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
    with torch.no_grad():
        b, s = head.decode_bboxes(cls, reg, IMG, cfg, rois=rois("cuda") if stage == "sr" else None)
    for i in range(2):
        assert_pool_matches_golden(b[i].cpu().numpy(), s[i].cpu().numpy(), G[f"{stage}_k{nms_pre}_boxes_{i}"],
                                   G[f"{stage}_k{nms_pre}_scores_{i}"], level_rows(A, nms_pre),
                                   f"{stage} {nms_pre} img {i} cl={channels_last}")


def test_decode_bboxes_goes_through_the_library(monkeypatch):
    """The arrays above come from r3det_level_pool, not from the torch form: count the library calls."""
    from r3det.ops import fr_boxes
    h0, _ = heads()
    cls, reg = maps("s0", "cuda")
    n = []
    real = fr_boxes.level_pool
    monkeypatch.setattr(fr_boxes, "level_pool", lambda *a, **k: (n.append(1), real(*a, **k))[1])
    with torch.no_grad():
        h0.cuda().decode_bboxes(cls, reg, IMG, dict(nms_pre=500))
    assert len(n) == 5


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("stage", ["s0", "sr"])
def test_get_bboxes_matches_reference_detections(stage, channels_last):
    """Pool + multiclass_nms_rotated (v1) of both heads against the reference's get_bboxes(with_nms=True)."""
    h0, hr = heads()
    head = (h0 if stage == "s0" else hr).cuda()
    cls, reg = maps(stage, "cuda", channels_last, untied=True)
    cfg = dict(nms_pre=500 if stage == "s0" else 100, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1),
               max_per_img=2000)
    res = head.get_bboxes(cls, reg, IMG, cfg, rois=rois("cuda") if stage == "sr" else None)
    for i, (d, lab) in enumerate(res):
        wd, wl = G[f"{stage}_dets_{i}"], G[f"{stage}_labels_{i}"]
        assert tuple(d.shape) == wd.shape and np.array_equal(lab.cpu().numpy(), wl), (stage, i, d.shape, wd.shape)
        d = d.cpu().numpy()
        assert np.abs(d[:, 5] - wd[:, 5]).max() <= 1e-6
        assert np.allclose(d[:, :5], wd[:, :5], rtol=1e-5, atol=1e-5)
