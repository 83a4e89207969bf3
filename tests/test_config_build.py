"""CPU: the ``model = dict(...)`` trees of the reference's shipped configs (tests/golden/configs.json, dumped from
configs/r3det/*.py and configs/rretinanet/*.py by tests/golden/make_golden_config.py) build this package's detectors
through ``build_detector`` -- the reference's registry names (R3Det, RRetinaNet, ResNet, FPN, RRetinaHead,
RRetinaRefineHead, RAnchorGenerator, PseudoAnchorGenerator, DeltaXYWHAOBBoxCoder, MaxIoUAssigner, FocalLoss,
SmoothL1Loss, L1Loss, RBboxOverlaps2D_v1/_v2/_v3) and constructor keywords (models/detectors/r3det.py:16-52,
dense_heads/rotate_retina_head.py:29-49, rotate_anchor_head.py:33-98) resolve unchanged."""
import json
import os

import pytest
import torch

from helpers import GOLDEN

CFG = json.load(open(os.path.join(GOLDEN, "configs.json")))
R3 = "configs/r3det/r3det_r50_fpn_1x_dota_v1.py"
TINY = "configs/r3det/r3det_tiny_r50_fpn_1x_dota_v1.py"
RR = "configs/rretinanet/rretinanet_obb_r50_fpn_1x_dota_v1.py"


def shapes(m):
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def test_r3det_config_builds_the_default_module_tree():
    from r3det.models import R3Det, build_detector
    from r3det.core.bbox.iou_calculators import RBboxOverlaps2D_v1
    m = build_detector(CFG[R3])
    assert isinstance(m, R3Det) and shapes(m) == shapes(R3Det())
    # the reference's state-dict names
    for k in ("backbone.layer1.0.conv1.weight", "neck.lateral_convs.0.conv.weight", "neck.fpn_convs.4.conv.bias",
              "bbox_head.cls_convs.3.conv.weight", "bbox_head.retina_reg.bias", "feat_refine_module.0.conv_5_1.weight",
              "refine_head.0.retina_cls.weight"):
        assert k in m.state_dict(), k
    assert m.num_refine_stages == 1 and len(m.feat_refine_module) == 1 and len(m.refine_head) == 1
    b, r = m.bbox_head, m.refine_head[0]
    assert (b.num_anchors, r.num_anchors) == (9, 1) and b.strides == r.strides == (8, 16, 32, 64, 128)
    assert b.assign_by_circumhbbox == 'v1' and r.assign_by_circumhbbox is None
    assert (b.assigner.pos_iou_thr, b.assigner.neg_iou_thr) == (0.5, 0.4)
    assert (r.assigner.pos_iou_thr, r.assigner.neg_iou_thr) == (0.6, 0.5)
    assert isinstance(b.assigner.iou_calculator, RBboxOverlaps2D_v1) and b.assigner.ignore_iof_thr == -1
    assert type(b.loss_cls).__name__ == 'FocalLoss' and (b.loss_cls.gamma, b.loss_cls.alpha) == (2.0, 0.25)
    assert type(b.loss_bbox).__name__ == 'SmoothL1Loss' and b.loss_bbox.beta == 0.11
    assert m.test_cfg == dict(nms_pre=2000, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
    assert b.test_cfg == m.test_cfg and r.test_cfg == m.test_cfg and m.train_cfg['stage_loss_weights'] == [1.0]
    assert m.feat_refine_module[0].featmap_strides == [8, 16, 32, 64, 128]
    assert m.backbone.frozen_stages == 1 and m.backbone.norm_eval
    # anchors of the generator the config names == the default head's
    a = b.anchor_generator.grid_priors([(4, 4), (2, 2), (1, 1), (1, 1), (1, 1)], device='cpu')
    d = R3Det().bbox_head.anchor_generator.grid_priors([(4, 4), (2, 2), (1, 1), (1, 1), (1, 1)], device='cpu')
    assert all(torch.equal(x, y) for x, y in zip(a, d)) and a[0].shape == (4 * 4 * 9, 5)


def test_tiny_and_rretinanet_configs():
    from r3det.models import RRetinaNet, build_detector
    from r3det.core.bbox.iou_calculators import RBboxOverlaps2D_v2, RBboxOverlaps2D_v3
    t = build_detector(CFG[TINY])
    assert len(t.bbox_head.cls_convs) == 2 and len(t.refine_head[0].reg_convs) == 2
    m = build_detector(CFG[RR])
    assert isinstance(m, RRetinaNet) and shapes(m) == shapes(RRetinaNet())
    assert m.bbox_head.assign_by_circumhbbox is None and type(m.bbox_head.loss_bbox).__name__ == 'L1Loss'
    assert type(RRetinaNet().bbox_head.loss_bbox).__name__ == 'L1Loss'
    h = build_detector(CFG["configs/rretinanet/rretinanet_hbb_r50_fpn_1x_dota_v1.py"])
    assert h.bbox_head.assign_by_circumhbbox == 'v1'
    # the v2 / v3 angle conventions select the other operator families through the same strings
    for ver, calc in (("v2", RBboxOverlaps2D_v2), ("v3", RBboxOverlaps2D_v3)):
        c = CFG[f"configs/rretinanet/rretinanet_obb_r50_fpn_1x_dota_{ver}.py"]
        m = build_detector(c)
        assert isinstance(m.bbox_head.assigner.iou_calculator, calc)
        assert m.test_cfg['nms'] == dict(type=ver, iou_thr=0.1) and m.bbox_head.bbox_coder.angle_range == ver


def test_unsupported_keywords_raise_instead_of_being_ignored():
    from r3det.models import build_detector
    import copy
    c = copy.deepcopy(CFG[R3])
    c['backbone']['depth'] = 101
    with pytest.raises(NotImplementedError):
        build_detector(c)
    c = copy.deepcopy(CFG[R3])
    c['bbox_head']['norm_cfg'] = dict(type='GN', num_groups=32)
    with pytest.raises(NotImplementedError):
        build_detector(c)
    c = copy.deepcopy(CFG[R3])
    c['bbox_head']['type'] = 'RetinaHead'
    with pytest.raises(KeyError):
        build_detector(c)
    c = copy.deepcopy(CFG[R3])
    c['refine_heads'] = []
    with pytest.raises(ValueError):
        build_detector(c)


def test_attribute_style_config_nodes_build_too():
    """mmcv.Config hands ConfigDict nodes (attribute access, keys()); anything with keys() is taken."""
    from r3det.models import build_detector

    class Node(dict):
        __getattr__ = dict.__getitem__

    def wrap(v):
        if isinstance(v, dict):
            return Node({k: wrap(x) for k, x in v.items()})
        return [wrap(x) for x in v] if isinstance(v, list) else v
    m = build_detector(wrap(CFG[RR]))
    assert m.bbox_head.num_anchors == 9 and isinstance(m.test_cfg, dict)
