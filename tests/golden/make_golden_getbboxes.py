"""Golden vectors for the pre-NMS pool and the detections of the two heads: run the REFERENCE's own

  r3det/models/dense_heads/rotate_anchor_head.py          RAnchorHead.get_bboxes / _get_bboxes_single
                                                          (:499-680: permute -> sigmoid -> max -> topk(nms_pre) ->
                                                          bbox_coder.decode(max_shape) -> cat -> background column)
  r3det/models/dense_heads/rotate_retina_refine_head.py   RRetinaRefineHead.get_bboxes (:147-200, rois as anchors)
  r3det/core/post_processing/bbox_nms_rotated.py          multiclass_nms_rotated (the with_nms=True half)

from where they lie, on CPU tensors in the build container, and record inputs + outputs
(tests/golden/getbboxes.npz: data only).  Third-party names are bound as in make_golden_heads.py
(mmdet stand-ins listed there); the compiled extensions behind the NMS wrappers are bound to oracle/_ref
(the reference's own CPU C++), as in make_golden_wrappers.py.

What the cases hold: five levels, ``nms_pre`` cutting two of them, tied best scores inside the
selected set, regression deltas beyond the coder's clip, centres decoded outside ``max_shape``, NCHW maps.
The best logit of every (position, anchor) row is a distinct value of a shuffled linspace, so that the top-k set and
its order do not depend on the last bit of sigmoid(); the explicit ties are exact float duplicates.

    python tests/golden/make_golden_getbboxes.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_heads as H  # noqa: E402  (binds nothing until bind_reference() is called)
from make_golden_heads import ANCHOR_CFG, CODER_CFG, LOSS_BBOX, LOSS_CLS, Cfg, O, _load, _mod, train_cfg  # noqa: E402


def bind_nms():
    """The reference's NMS wrappers + multiclass_nms_rotated on top of the modules bind_reference() made,
    extension names bound to oracle/_ref."""
    def rnms(dets, thr):
        return torch.from_numpy(O.ref_v1_rnms(dets.numpy(), thr))

    def nms_rotated(b, s, thr):
        return torch.from_numpy(O.ref_v3_nms(b.contiguous().numpy(), s.contiguous().numpy(), thr))

    _mod("r3det.ops.rnms", True)
    _mod("r3det.ops.rnms.rnms_ext", rnms=rnms)
    w1 = _load("r3det.ops.rnms.rnms_wrapper", "r3det/ops/rnms/rnms_wrapper.py")
    _mod("r3det.ops.nms_rotated", True)
    _mod("r3det.ops.nms_rotated.nms_rotated_ext", nms_rotated=nms_rotated)
    w3 = _load("r3det.ops.nms_rotated.nms_rotated_wrapper", "r3det/ops/nms_rotated/nms_rotated_wrapper.py")
    ops = sys.modules["r3det.ops"]
    ops.batched_rnms, ops.rnms = w1.batched_rnms, w1.rnms
    ops.obb_batched_nms, ops.obb_nms = w3.obb_batched_nms, w3.obb_nms
    _mod("r3det.core.post_processing", True)
    pp = _load("r3det.core.post_processing.bbox_nms_rotated", "r3det/core/post_processing/bbox_nms_rotated.py")
    sys.modules["r3det.models.dense_heads.rotate_anchor_head"].multiclass_nms_rotated = pp.multiclass_nms_rotated


def spread_cls(N, A, C, h, w, g, lo=-6.0, hi=4.0):
    """(N, A*C, h, w) logits whose best class per row (p, a) is a distinct value of a shuffled linspace."""
    L = h * w * A
    cls = torch.empty(N, L, C)
    for n in range(N):
        best = torch.linspace(lo, hi, L)[torch.randperm(L, generator=g)]
        rest = best[:, None] - 0.5 - 3 * torch.rand(L, C, generator=g)
        rest[torch.arange(L), torch.randint(0, C, (L,), generator=g)] = best
        cls[n] = rest
    return cls.view(N, h, w, A * C).permute(0, 3, 1, 2).contiguous()


def tie_rows(cls, n, A, C, rows, value, out, key):
    """Give the rows (p * A + a) of image n the same best logit ``value`` (an exact tie inside the top-k set).
    The rows' original values are recorded under ``key`` so that a test can rebuild the tie-free map (the detections
    are recorded on the tie-free maps: with tied scores the NMS order is open)."""
    N, _, h, w = cls.shape
    v = cls.permute(0, 2, 3, 1).reshape(N, h * w * A, C)  # a copy (permute of a contiguous NCHW tensor)
    out[key + "_img"], out[key + "_rows"] = np.array(n), np.array(rows)
    out[key + "_values"] = v[n, rows].numpy().copy()
    for r in rows:
        v[n, r] = value - 1.0
        v[n, r, r % C] = value
    return v.view(N, h, w, A * C).permute(0, 3, 1, 2).contiguous()


def main():
    assert os.path.isdir(H.REF)
    R = H.bind_reference()
    bind_nms()
    out = {}
    g = torch.Generator().manual_seed(11)
    N, C = 2, 15
    img = (256, 192)
    sizes = [(32, 24), (16, 12), (8, 6), (4, 3), (2, 2)]
    out["sizes"], out["img_shape"] = np.array(sizes), np.array(img)
    metas = [dict(img_shape=img + (3,), pad_shape=img + (3,), scale_factor=1.0) for _ in range(N)]

    # ---- first stage: 9 anchors per position, nms_pre cuts levels 0 (6912 rows) and 1 (1728 rows)
    A = 9
    head = R.RRetinaHead(num_classes=C, in_channels=8, stacked_convs=1, feat_channels=8, anchor_generator=ANCHOR_CFG,
                         bbox_coder=CODER_CFG, loss_cls=LOSS_CLS, loss_bbox=LOSS_BBOX, train_cfg=train_cfg(0.5, 0.4))
    cls = [spread_cls(N, A, C, h, w, g) for h, w in sizes]
    cls_untied = list(cls)
    cls[0] = tie_rows(cls[0], 0, A, C, [17, 4000, 4001, 6911], 5.0, out, "s0_untie_l0")  # four-way tie above everything
    cls[1] = tie_rows(cls[1], 1, A, C, [3, 1700], 3.9990234375, out, "s0_untie_l1")      # a two-way tie inside the set
    # deltas: dx / dy up to several box sizes (centres leave the image -> max_shape clamp), dw / dh beyond
    # the clip |d| <= log(1000 / 16) = 4.135
    reg = [torch.randn(N, A * 5, h, w, generator=g) * torch.tensor([1.5, 1.5, 3.0, 3.0, 0.7]).repeat(A)[None, :, None, None]
           for h, w in sizes]
    for l in range(5):
        out[f"s0_cls_l{l}"], out[f"s0_reg_l{l}"] = cls[l].numpy(), reg[l].numpy()
    for nms_pre in (500, -1):
        cfg = Cfg(nms_pre=nms_pre, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
        res = head.get_bboxes(cls, reg, metas, cfg, with_nms=False)
        for i, (b, s) in enumerate(res):
            out[f"s0_k{nms_pre}_boxes_{i}"], out[f"s0_k{nms_pre}_scores_{i}"] = b.numpy(), s.numpy()
        print("stage 0 nms_pre", nms_pre, "pool", tuple(res[0][0].shape), tuple(res[0][1].shape))
    cfg = Cfg(nms_pre=500, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
    # detections on the tie-free maps, with the pool they were made from (the boxes of a pool are reproduced to
    # 1e-5 only by another implementation of exp / sin; a dense pool's NMS is not stable under that, so the
    # composition pool -> multiclass_nms_rotated is pinned on the reference's own pool bits)
    for i, (b, s) in enumerate(head.get_bboxes(cls_untied, reg, metas, cfg, with_nms=False)):
        out[f"s0_untied_boxes_{i}"], out[f"s0_untied_scores_{i}"] = b.numpy(), s.numpy()
    for i, (d, lab) in enumerate(head.get_bboxes(cls_untied, reg, metas, cfg)):
        out[f"s0_dets_{i}"], out[f"s0_labels_{i}"] = d.numpy(), lab.numpy()
        print("stage 0 detections", i, tuple(d.shape))

    # ---- refine stage: the rois of filter_bboxes are the anchors, one per position; nms_pre cuts levels 0 and 1
    small = [torch.randn(N, A * 5, h, w, generator=g) * torch.tensor([0.3, 0.3, 0.6, 0.6, 0.4]).repeat(A)[None, :, None, None]
             for h, w in sizes]
    rois = head.filter_bboxes(cls, small)
    rhead = R.RRetinaRefineHead(num_classes=C, in_channels=8, stacked_convs=1, feat_channels=8,
                                assign_by_circumhbbox=None, bbox_coder=CODER_CFG, loss_cls=LOSS_CLS,
                                loss_bbox=LOSS_BBOX, train_cfg=train_cfg(0.6, 0.5))
    rcls = [spread_cls(N, 1, C, h, w, g) for h, w in sizes]
    rcls_untied = list(rcls)
    rcls[0] = tie_rows(rcls[0], 1, 1, C, [5, 6, 700], 5.0, out, "sr_untie_l0")
    rreg = [torch.randn(N, 5, h, w, generator=g) * torch.tensor([1.0, 1.0, 2.5, 2.5, 0.5])[None, :, None, None]
            for h, w in sizes]
    for l in range(5):
        out[f"sr_cls_l{l}"], out[f"sr_reg_l{l}"] = rcls[l].numpy(), rreg[l].numpy()
        for i in range(N):
            out[f"sr_rois_{i}_l{l}"] = rois[i][l].numpy()
    # the with_nms=False half of RRetinaRefineHead.get_bboxes = _get_bboxes_single on (maps of image i, rois[i])
    for nms_pre in (100, -1):
        cfg = Cfg(nms_pre=nms_pre, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
        for i in range(N):
            b, s = rhead._get_bboxes_single([c[i] for c in rcls], [r[i] for r in rreg], rois[i], metas[i]['img_shape'],
                                            1.0, cfg, False, with_nms=False)
            out[f"sr_k{nms_pre}_boxes_{i}"], out[f"sr_k{nms_pre}_scores_{i}"] = b.numpy(), s.numpy()
        print("refine nms_pre", nms_pre, "pool", tuple(b.shape), tuple(s.shape))
    cfg = Cfg(nms_pre=100, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
    for i in range(N):
        b, s = rhead._get_bboxes_single([c[i] for c in rcls_untied], [r[i] for r in rreg], rois[i], metas[i]['img_shape'],
                                        1.0, cfg, False, with_nms=False)
        out[f"sr_untied_boxes_{i}"], out[f"sr_untied_scores_{i}"] = b.numpy(), s.numpy()
    for i, (d, lab) in enumerate(rhead.get_bboxes(rcls_untied, rreg, metas, cfg, rois=rois)):
        out[f"sr_dets_{i}"], out[f"sr_labels_{i}"] = d.numpy(), lab.numpy()
        print("refine detections", i, tuple(d.shape))
    np.savez_compressed(os.path.join(HERE, "getbboxes.npz"), **out)
    print("done", len(out), "arrays")


if __name__ == "__main__":
    main()
