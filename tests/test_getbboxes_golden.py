"""CPU: the op-by-op form of the pre-NMS pool (RRetinaHead.decode_bboxes_torch) against the arrays recorded from
the REFERENCE's own RAnchorHead._get_bboxes_single(with_nms=False) and RRetinaRefineHead.get_bboxes
(rotate_anchor_head.py:590-680, rotate_retina_refine_head.py:147-200; tests/golden/getbboxes.npz, written by
tests/golden/make_golden_getbboxes.py).  The fused library call is compared with the same arrays in
tests/test_gpu_getbboxes_golden.py."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_pool_matches_golden

G = np.load(os.path.join(GOLDEN, "getbboxes.npz"))
SIZES = [tuple(int(v) for v in s) for s in G["sizes"]]
IMG = tuple(int(v) for v in G["img_shape"])


def heads():
    from r3det.models.heads import RRetinaHead, RRetinaRefineHead
    torch.manual_seed(0)
    return (RRetinaHead(num_classes=15, in_channels=8, feat_channels=8, stacked_convs=1).eval(),
            RRetinaRefineHead(num_classes=15, in_channels=8, feat_channels=8, stacked_convs=1).eval())


def maps(prefix, dev="cpu", channels_last=False, untied=False):
    """The recorded head maps; ``untied``: with the rows that were overwritten to make exact score ties restored
    (the detections were recorded on those: with tied scores the NMS order is open)."""
    out = []
    for kind in ("cls", "reg"):
        ts = [torch.from_numpy(G[f"{prefix}_{kind}_l{l}"]).clone() for l in range(5)]
        for l in range(5):
            if untied and kind == "cls" and f"{prefix}_untie_l{l}_rows" in G:
                N, AC, h, w = ts[l].shape
                C = G[f"{prefix}_untie_l{l}_values"].shape[1]
                v = ts[l].permute(0, 2, 3, 1).reshape(N, -1, C)
                v[int(G[f"{prefix}_untie_l{l}_img"]), torch.from_numpy(G[f"{prefix}_untie_l{l}_rows"])] = \
                    torch.from_numpy(G[f"{prefix}_untie_l{l}_values"])
                ts[l] = v.view(N, h, w, AC).permute(0, 3, 1, 2).contiguous()
        ts = [t.to(dev) for t in ts]
        if channels_last:
            ts = [t.contiguous(memory_format=torch.channels_last) for t in ts]
        out.append(ts)
    return out


def rois(dev="cpu"):
    return [[torch.from_numpy(G[f"sr_rois_{i}_l{l}"]).to(dev) for l in range(5)] for i in range(2)]


def level_rows(A, nms_pre):
    rows = [h * w * A for h, w in SIZES]
    return [min(r, nms_pre) if nms_pre > 0 else r for r in rows]


@pytest.mark.parametrize("stage,nms_pre", [("s0", 500), ("s0", -1), ("sr", 100), ("sr", -1)])
def test_torch_form_matches_reference_pool(stage, nms_pre):
    h0, hr = heads()
    head, A = (h0, 9) if stage == "s0" else (hr, 1)
    cls, reg = maps(stage)
    cfg = dict(nms_pre=nms_pre, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
    with torch.no_grad():
        b, s = head.decode_bboxes_torch(cls, reg, IMG, cfg, rois=rois() if stage == "sr" else None)
    for i in range(2):
        assert_pool_matches_golden(b[i].numpy(), s[i].numpy(), G[f"{stage}_k{nms_pre}_boxes_{i}"],
                                   G[f"{stage}_k{nms_pre}_scores_{i}"], level_rows(A, nms_pre), f"{stage} {nms_pre} img {i}")


def test_golden_holds_the_cases_it_claims():
    """The recorded pools contain what the fixture was built for: cut levels, ties inside the selected set, clipped
    sizes and clamped centres."""
    b, s = G["s0_k500_boxes_0"], G["s0_k500_scores_0"]
    assert b.shape == (500 + 500 + 432 + 108 + 36, 5)
    best = s[:500, :-1].max(1)
    assert (best[:4] == best[0]).all() and best[4] < best[0]          # the four-way tie leads level 0
    assert (np.diff(best) <= 0).all()
    b1 = G["s0_k500_scores_1"][500:1000, :-1].max(1)
    assert (np.diff(b1) == 0).sum() == 1                                 # the two-way tie inside level 1
    H, W = IMG
    assert (b[:, 0] == 0).any() and (b[:, 0] == W - 1).any() and (b[:, 1] == 0).any() and (b[:, 1] == H - 1).any()
    a = G["s0_k-1_boxes_0"]
    assert a.shape[0] == sum(h * w * 9 for h, w in SIZES)
    r = G["sr_k100_boxes_1"]
    assert r.shape == (100 + 100 + 48 + 12 + 4, 5)
    rb = G["sr_k100_scores_1"][:100, :-1].max(1)
    assert (rb[:3] == rb[0]).all()
