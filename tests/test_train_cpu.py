"""CPU: the training glue (targets, losses, detector wiring) of BASELINE configs[4].

(1) ``RRetinaHead.filter_bboxes`` / ``RRetinaRefineHead.refine_bboxes`` (torch forms) and the heads'
``loss`` / ``get_targets`` against outputs of the REFERENCE's own head code (tests/golden/heads.npz,
made by tests/golden/make_golden_heads.py: reference glue, third-party pieces stood in).
(2) ``R3Det.forward_train`` end to end on a tiny input: keys, finiteness, every trainable parameter
gets a gradient, FRM gradient flows.
The three HIP kernels of the step are replaced by TEST-ONLY CPU stand-ins (tests/cpu_standins.py)."""
import os

import numpy as np
import pytest
import torch

from cpu_standins import cpu_kernels
from helpers import GOLDEN

G = np.load(os.path.join(GOLDEN, "heads.npz"))


def t(name):
    return torch.from_numpy(G[name])


def metas(tag, n=2):
    H, W = {"a": (128, 96), "b": (64, 64)}[tag]
    return [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0) for _ in range(n)]


def load_case(tag):
    cls = [t(f"{tag}_cls_l{l}") for l in range(5)]
    reg = [t(f"{tag}_reg_l{l}") for l in range(5)]
    gtb = [t(f"{tag}_gt_bboxes_{i}") for i in range(2)]
    gtl = [t(f"{tag}_gt_labels_{i}") for i in range(2)]
    return cls, reg, gtb, gtl


@pytest.mark.parametrize("tag", ["a", "b"])
def test_filter_and_refine_bboxes_torch_vs_reference(tag):
    from r3det.models.heads import RRetinaHead, RRetinaRefineHead
    cls, reg, _, _ = load_case(tag)
    head = RRetinaHead()
    rois = head.filter_bboxes(cls, reg)
    for i in range(2):
        for l in range(5):
            want = G[f"{tag}_rois_{i}_l{l}"]
            assert rois[i][l].shape == want.shape
            assert np.allclose(rois[i][l].numpy(), want, rtol=1e-6, atol=1e-5), (i, l)
    rhead = RRetinaRefineHead()
    rcls = [t(f"{tag}_rcls_l{l}") for l in range(5)]
    rreg = [t(f"{tag}_rreg_l{l}") for l in range(5)]
    ref_rois = [[t(f"{tag}_rois_{i}_l{l}") for l in range(5)] for i in range(2)]
    refined = rhead.refine_bboxes(rcls, rreg, ref_rois)
    for i in range(2):
        for l in range(5):
            assert np.allclose(refined[i][l].numpy(), G[f"{tag}_refined_{i}_l{l}"], rtol=1e-6, atol=1e-5), (i, l)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_head_losses_and_targets_vs_reference_glue(tag):
    """Targets exactly (labels, weights) / to rounding (encoded deltas); per-level losses to 1e-5 rel."""
    from r3det.models.heads import RRetinaHead, RRetinaRefineHead
    cls, reg, gtb, gtl = load_case(tag)
    head = RRetinaHead()
    with cpu_kernels():
        sizes = [c.shape[-2:] for c in cls]
        anchor_list, flags = head.get_anchors(sizes, metas(tag), 'cpu')
        assert flags == [None, None]
        labels, lw, bt, bw, npos = head.get_targets(anchor_list, flags, gtb, metas(tag), gtl)
        assert int(npos) == int(G[f"{tag}_s0_num_total_pos"])
        for l in range(5):
            assert np.array_equal(labels[l].numpy(), G[f"{tag}_s0_labels_l{l}"])
            assert np.array_equal(lw[l].numpy(), G[f"{tag}_s0_label_weights_l{l}"])
            assert np.array_equal(bw[l].numpy(), G[f"{tag}_s0_bbox_weights_l{l}"])
            assert np.allclose(bt[l].numpy(), G[f"{tag}_s0_bbox_targets_l{l}"], rtol=1e-6, atol=1e-6)
        losses = head.loss(cls, reg, gtb, gtl, metas(tag))
        got_c = np.array([float(v) for v in losses['loss_cls']])
        got_b = np.array([float(v) for v in losses['loss_bbox']])
        assert np.allclose(got_c, G[f"{tag}_s0_loss_cls"], rtol=1e-5, atol=1e-7)
        assert np.allclose(got_b, G[f"{tag}_s0_loss_bbox"], rtol=1e-5, atol=1e-7)
        rhead = RRetinaRefineHead()
        rcls = [t(f"{tag}_rcls_l{l}") for l in range(5)]
        rreg = [t(f"{tag}_rreg_l{l}") for l in range(5)]
        rois = [[t(f"{tag}_rois_{i}_l{l}") for l in range(5)] for i in range(2)]
        rl = rhead.loss(rcls, rreg, gtb, gtl, metas(tag), rois=rois)
        assert np.allclose([float(v) for v in rl['loss_cls']], G[f"{tag}_sr_loss_cls"], rtol=1e-5, atol=1e-7)
        assert np.allclose([float(v) for v in rl['loss_bbox']], G[f"{tag}_sr_loss_bbox"], rtol=1e-5, atol=1e-7)


def test_targets_with_partly_invalid_anchors_match_index_form():
    """pad_shape smaller than the feature grid: anchors outside are left out of the assignment and
    come back as background with weight 0 (rotate_anchor_head.py:203-207,262-272)."""
    from r3det.models.heads import RRetinaHead
    cls, reg, gtb, gtl = load_case("a")
    head = RRetinaHead()
    m = [dict(img_shape=(100, 90, 3), pad_shape=(100, 90, 3), scale_factor=1.0) for _ in range(2)]
    with cpu_kernels():
        sizes = [c.shape[-2:] for c in cls]
        anchor_list, flags = head.get_anchors(sizes, m, 'cpu')
        assert flags[0] is not None and not bool(flags[0].all())
        labels, lw, bt, bw, npos = head.get_targets(anchor_list, flags, gtb, m, gtl)
        flat = torch.cat(anchor_list[0])
        inside = flags[0]
        sub = head._targets_single(flat[inside], None, gtb[0], gtl[0], m[0])
        full_labels = torch.cat([x[0] for x in labels])
        assert torch.equal(full_labels[inside], sub[0])
        assert bool((full_labels[~inside] == head.num_classes).all())
        assert float(torch.cat([x[0] for x in lw])[~inside].abs().max()) == 0


def tiny_batch(seed, n=2, size=64, n_gt=6):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(n, 3, size, size, generator=g)
    gtb, gtl = [], []
    for i in range(n):
        u = torch.rand(n_gt, 5, generator=g)
        gtb.append(torch.stack([u[:, 0] * size, u[:, 1] * size, 6 + u[:, 2] * 30, 6 + u[:, 3] * 20,
                                -u[:, 4] * (np.pi / 2)], 1))
        gtl.append(torch.randint(0, 15, (n_gt,), generator=g))
    return img, gtb, gtl


def test_forward_train_tiny_end_to_end():
    from r3det.models import R3Det
    from r3det.models.detectors import parse_losses
    torch.manual_seed(0)
    model = R3Det().train()
    assert not model.backbone.bn1.training and not model.backbone.conv1.weight.requires_grad  # norm_eval, frozen stem
    img, gtb, gtl = tiny_batch(1)
    with cpu_kernels():
        losses = model(img, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
        assert sorted(losses) == ['s0.loss_bbox', 's0.loss_cls', 'sr0.loss_bbox', 'sr0.loss_cls']
        assert all(len(v) == 5 for v in losses.values())
        loss, log_vars = parse_losses(losses)
        assert torch.isfinite(loss) and float(loss.detach()) > 0
        assert float(log_vars['loss']) == pytest.approx(sum(float(v) for k, v in log_vars.items() if k != 'loss'),
                                                        rel=1e-5)
        loss.backward()
    missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert missing == []
    frm = model.feat_refine_module[0]
    assert float(frm.conv_1_1.weight.grad.abs().sum()) > 0 and float(frm.conv_5_1.weight.grad.abs().sum()) > 0
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_train_step_and_stage_loss_weights():
    from r3det.models import R3Det
    from r3det.models.detectors import R3DET_TRAIN_CFG
    torch.manual_seed(0)
    cfg = dict(R3DET_TRAIN_CFG, stage_loss_weights=[0.5])
    m1, m2 = R3Det().train(), R3Det(train_cfg=cfg).train()
    m2.load_state_dict(m1.state_dict())
    img, gtb, gtl = tiny_batch(2)
    with cpu_kernels():
        o1 = m1.train_step(dict(img=img, gt_bboxes=gtb, gt_labels=gtl))
        o2 = m2.train_step(dict(img=img, gt_bboxes=gtb, gt_labels=gtl))
    assert o1['num_samples'] == 2
    for k in ('s0.loss_cls', 's0.loss_bbox'):
        assert float(o1['log_vars'][k]) == pytest.approx(float(o2['log_vars'][k]), rel=1e-6)
    for k in ('sr0.loss_cls', 'sr0.loss_bbox'):
        assert float(o2['log_vars'][k]) == pytest.approx(0.5 * float(o1['log_vars'][k]), rel=1e-6)


def test_rretinanet_forward_train_assigns_on_oriented_gt():
    from r3det.models import RRetinaNet
    torch.manual_seed(0)
    m = RRetinaNet().train()
    assert m.bbox_head.assign_by_circumhbbox is None
    img, gtb, gtl = tiny_batch(3)
    with cpu_kernels():
        losses = m(img, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
    assert sorted(losses) == ['loss_bbox', 'loss_cls'] and all(torch.isfinite(v) for v in losses['loss_cls'])
