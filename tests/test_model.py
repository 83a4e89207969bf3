"""Host-side model glue (r3det/models): coder, anchors, best-anchor selection and the batched
get_bboxes against a per-image / per-level restatement of the reference's loops
(rotate_retina_head.py:117-179, rotate_anchor_head.py:590-675)."""
import math

import pytest
import torch

from r3det.models.coder import bbox2delta_v1, delta2bbox_v1
from r3det.models.heads import RRetinaHead, RRetinaRefineHead, level_anchors
from r3det import synthetic as syn


def test_coder_roundtrip_and_clamps():
    g = torch.Generator().manual_seed(0)
    rois = syn.rand_rboxes(200, 1)
    gt = syn.rand_rboxes(200, 2)
    d = bbox2delta_v1(rois, gt)
    back = delta2bbox_v1(rois, d)
    assert torch.allclose(back, gt, rtol=1e-4, atol=1e-3)
    # dw/dh are clamped to |log(16/1000)|, centres to the image when max_shape is given
    big = torch.zeros(1, 5)
    big[0, 2] = 100.0
    out = delta2bbox_v1(torch.tensor([[10., 10., 4., 4., 0.]]), big)
    assert out[0, 2].item() == pytest.approx(4 * math.exp(abs(math.log(16 / 1000))), rel=1e-6)
    out = delta2bbox_v1(torch.tensor([[10., 10., 4., 4., 0.]]), torch.tensor([[-100., 500., 0., 0., 0.3]]),
                        max_shape=(64, 128))
    assert out[0, 0].item() == 0 and out[0, 1].item() == 63 and out[0, 4].item() == pytest.approx(0.3)


def test_anchor_order_matches_grid_helper():
    per_level = [level_anchors((1024 // s, 1024 // s), s, 'cpu') for s in syn.STRIDES]
    assert [a.shape[0] for a in per_level] == [147456, 36864, 9216, 2304, 576]
    assert torch.equal(torch.cat(per_level), syn.anchor_grid())
    # an independent float64 statement of the same grid (tests/helpers.py); the generator's fp32 corner
    # arithmetic (x +- w/2, then their difference, as mmdet does) rounds widths by up to ~1e-4
    from helpers import anchor_grid as np_grid
    assert torch.allclose(torch.cat(per_level), torch.from_numpy(np_grid()), rtol=0, atol=2e-4)
    a0 = per_level[0][:9]
    # ratio-major, scale-minor: first three anchors share ratio 1 (w == h), scales 4, 5.04, 6.35 x stride
    assert torch.allclose(a0[:3, 2], a0[:3, 3])
    assert a0[0, 2].item() == pytest.approx(32.0) and a0[2, 2].item() == pytest.approx(8 * 4 * 2 ** (2 / 3), rel=1e-6)
    assert (a0[3, 2] > a0[3, 3]) and (a0[6, 2] < a0[6, 3])  # ratio 0.5 = h/w, then ratio 2
    assert (per_level[0][9, :2] == torch.tensor([8., 0.])).all()  # x runs fastest


def _loop_filter(head, cls_scores, bbox_preds):
    """Per-image restatement of filter_bboxes."""
    N = cls_scores[0].size(0)
    out = [[] for _ in range(N)]
    for lvl, (cls, reg) in enumerate(zip(cls_scores, bbox_preds)):
        anc = level_anchors(cls.shape[-2:], head.strides[lvl], cls.device).reshape(-1, 9, 5)
        for i in range(N):
            c = cls[i].permute(1, 2, 0).reshape(-1, 9, head.num_classes)
            best = c.max(-1)[0].argmax(-1)
            r = reg[i].permute(1, 2, 0).reshape(-1, 9, 5)
            idx = torch.arange(r.size(0))
            out[i].append(delta2bbox_v1(anc[idx, best], r[idx, best]))
    return out


def test_filter_bboxes_matches_loop():
    torch.manual_seed(3)
    head = RRetinaHead(num_classes=15, in_channels=8, feat_channels=8, strides=(8, 16))
    feats = [torch.randn(2, 8, 8, 8), torch.randn(2, 8, 4, 4)]
    with torch.no_grad():
        cls, reg = head(feats)
        cls = [c + torch.randn_like(c) for c in cls]
        reg = [r + 0.2 * torch.randn_like(r) for r in reg]
    got = head.filter_bboxes(cls, reg)
    want = _loop_filter(head, cls, reg)
    for i in range(2):
        for l in range(2):
            assert got[i][l].shape == (feats[l].shape[-1] ** 2, 5)
            assert torch.equal(got[i][l], want[i][l])


def test_state_dict_names_and_param_count():
    from r3det.models import R3Det
    m = R3Det()
    keys = m.state_dict().keys()
    for k in ["backbone.layer1.0.conv1.weight", "backbone.layer4.2.bn3.running_var", "neck.lateral_convs.0.conv.weight",
              "neck.fpn_convs.4.conv.bias", "bbox_head.cls_convs.3.conv.weight", "bbox_head.retina_reg.bias",
              "feat_refine_module.0.conv_5_1.weight", "refine_head.0.retina_cls.weight"]:
        assert k in keys, k
    assert m.bbox_head.retina_cls.out_channels == 9 * 15 and m.refine_head[0].retina_cls.out_channels == 15
    n = sum(p.numel() for p in m.parameters())
    assert 41e6 < n < 43e6  # ~42 M parameters
    assert not m.backbone.layer1[0].conv1.weight.requires_grad  # frozen_stages=1
    assert m.backbone.layer2[0].conv1.weight.requires_grad
    assert m.bbox_head.retina_cls.bias[0].item() == pytest.approx(-math.log(99), rel=1e-6)


def test_models_copy_and_pickle():
    """the heads own a CapacityHint (a lock inside); deepcopy (EMA / SWA twins), torch.save of the whole
    module and pickling must work, and the copy starts with an empty hint of its own."""
    import copy
    import io
    import pickle
    from r3det.models import R3Det
    m = R3Det()
    m.bbox_head.nms_hint.put(("k",), 7)
    c = copy.deepcopy(m)
    assert c.bbox_head.nms_hint is not m.bbox_head.nms_hint and c.bbox_head.nms_hint.get(("k",)) is None
    assert m.bbox_head.nms_hint.get(("k",)) == 7
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), c.state_dict().values()))
    head = pickle.loads(pickle.dumps(m.refine_head[0]))
    assert head.nms_hint.get(("k",)) is None
    head.nms_hint.put(("a",), 1)  # the recreated lock works
    buf = io.BytesIO()
    torch.save(m.bbox_head, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)
    assert again.retina_cls.weight.shape == m.bbox_head.retina_cls.weight.shape


def _loop_get_bboxes(head, cls_scores, bbox_preds, img_shape, cfg, rois):
    """Per-image, per-level restatement (rotate_anchor_head.py:590-675)."""
    from r3det.core.post_processing import multiclass_nms_rotated
    N = cls_scores[0].size(0)
    res = []
    for i in range(N):
        mb, ms = [], []
        for l, (cls, reg) in enumerate(zip(cls_scores, bbox_preds)):
            scores = cls[i].permute(1, 2, 0).reshape(-1, head.cls_out_channels).sigmoid()
            pred = reg[i].permute(1, 2, 0).reshape(-1, 5)
            anc = rois[i][l] if rois is not None else level_anchors(cls.shape[-2:], head.strides[l], cls.device)
            if 0 < cfg['nms_pre'] < scores.shape[0]:
                top = scores.max(1)[0].topk(cfg['nms_pre'])[1]
                anc, pred, scores = anc[top], pred[top], scores[top]
            mb.append(delta2bbox_v1(anc, pred, max_shape=img_shape))
            ms.append(scores)
        mb, ms = torch.cat(mb), torch.cat(ms)
        ms = torch.cat([ms, ms.new_zeros(ms.shape[0], 1)], 1)
        res.append(multiclass_nms_rotated(mb, ms, cfg['score_thr'], cfg['nms'], cfg['max_per_img']))
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("refine", [False, True])
def test_batched_get_bboxes_matches_loop(refine):
    torch.manual_seed(5)
    dev = torch.device('cuda')
    cfg = dict(nms_pre=50, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
    head = (RRetinaRefineHead if refine else RRetinaHead)(num_classes=15, in_channels=8, feat_channels=8,
                                                           strides=(8, 16, 32), test_cfg=cfg).to(dev)
    sizes = [16, 8, 4]
    N = 3
    cls = [torch.randn(N, head.num_anchors * 15, s, s, device=dev) * 2 - 3 for s in sizes]
    reg = [torch.randn(N, head.num_anchors * 5, s, s, device=dev) * 0.3 for s in sizes]
    rois = None
    if refine:
        rois = [[syn.fr_level_boxes(1, s, s, st, 10 * i + j, device=dev) for j, (s, st) in enumerate(zip(sizes, (8, 16, 32)))]
                for i in range(N)]
    got = head.get_bboxes(cls, reg, (128, 128), cfg, rois=rois)
    want = _loop_get_bboxes(head, cls, reg, (128, 128), cfg, rois)
    assert len(got) == N
    for (gd, gl), (wd, wl) in zip(got, want):
        assert gd.shape[0] > 0
        assert torch.equal(gd, wd) and torch.equal(gl, wl)


@pytest.mark.gpu
def test_r3det_end_to_end_small():
    """Full R3Det simple_test on 2 x 256^2 tiles: runs through FRM + refine head + NMS, output
    format of the reference (dets (k,6), labels (k,)), deterministic, keep lists consistent with
    a second NMS pass."""
    from r3det.models import R3Det, RRetinaNet
    from r3det.models.detectors import calibrate_score_bias
    from r3det.ops import batched_rnms
    torch.manual_seed(7)
    dev = torch.device('cuda')
    m = R3Det().eval().to(dev)
    img = torch.randn(2, 3, 256, 256, device=dev)
    calibrate_score_bias(m, img, frac=0.02)
    res = m.simple_test(img)
    assert len(res) == 2
    # determinism of the custom-op part on fixed conv outputs (MIOpen convs themselves may pick
    # non-deterministic algorithms between runs, so the whole model is not compared bitwise)
    with torch.no_grad():
        x = m.extract_feat(img)
        cls, reg = m.bbox_head(x)
        rois = m.bbox_head.filter_bboxes(cls, reg)
        xr = m.feat_refine_module[0](x, rois)
        cls, reg = m.refine_head[0](xr)
        res = m.refine_head[0].get_bboxes(cls, reg, img.shape[-2:], m.test_cfg, rois=rois)
        res2 = m.refine_head[0].get_bboxes(cls, reg, img.shape[-2:], m.test_cfg, rois=rois)
    for (d, l), (d2, l2) in zip(res, res2):
        assert d.dim() == 2 and d.size(1) == 6 and l.shape == (d.size(0),) and l.dtype == torch.long
        assert 0 < d.size(0) <= 2000 and (d[:, 5] > 0.05).all() and l.min() >= 0 and l.max() < 15
        assert torch.equal(d, d2) and torch.equal(l, l2)
        # idempotence: NMS of the kept set keeps everything
        kept, keep = batched_rnms(d[:, :5].contiguous(), d[:, 5].contiguous(), l, 0.1)
        assert keep.numel() == d.size(0)
    r = RRetinaNet().eval().to(dev)
    calibrate_score_bias(r, img, frac=0.005)
    out = r.simple_test(img)
    assert len(out) == 2 and out[0][0].size(1) == 6


def _randomise_bn(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)


def test_fuse_conv_bn_and_epilogues_cpu():
    """BatchNorm folded into the convolutions (the reference benchmark's --fuse-conv-bn) and the
    one-pass epilogues (torch fallback on CPU) reproduce the eval-mode network up to rounding."""
    from r3det.models.backbone import FPN, ResNet50
    from r3det.models.fuse import fuse_conv_bn, fuse_epilogues
    torch.manual_seed(0)
    net = torch.nn.Sequential(ResNet50(), FPN()).eval()
    _randomise_bn(net)
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        want = net(x)
        fuse_conv_bn(net)
        assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in net.modules())
        got1 = net(x)
        fuse_epilogues(net)
        got2 = net(x)
    for w, a, b in zip(want, got1, got2):
        scale = w.abs().max().item()
        assert (a - w).abs().max().item() <= 1e-5 * scale
        assert (b - w).abs().max().item() <= 1e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("channels_last", [False, True])
def test_fused_inference_model_gpu(channels_last):
    """r3det_bias_act (NCHW and channels_last) inside the fused model: same feature maps up to rounding."""
    from r3det.models.backbone import FPN, ResNet50
    from r3det.models.fuse import fuse_for_inference
    from r3det.ops.epilogue import bias_act_
    torch.manual_seed(0)
    net = torch.nn.Sequential(ResNet50(), FPN()).eval()
    _randomise_bn(net)
    net = net.cuda()
    x = torch.randn(2, 3, 256, 192, device='cuda')
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    with torch.no_grad():
        want = net(x)
        fuse_for_inference(net)
        net = net.to(memory_format=fmt)
        got = net(x.contiguous(memory_format=fmt))
    for w, a in zip(want, got):
        assert (a - w).abs().max().item() <= 2e-5 * w.abs().max().item()
    # the op itself, against its definition, exactly (adds in the same order)
    # (the last two: more float4 than the capped grid has threads -- several trips per thread, the grid stride a
    # multiple of the channel count (64) and not (12))
    for shape in [(2, 8, 5, 7), (3, 64, 16, 16), (1, 256, 1, 1), (2, 64, 384, 384), (4, 12, 512, 512)]:
        y = torch.randn(shape, device='cuda').contiguous(memory_format=fmt)
        b = torch.randn(shape[1], device='cuda')
        r = torch.randn(shape, device='cuda').contiguous(memory_format=fmt)
        for res in (None, r):
            for relu in (False, True):
                ref = y + b.view(1, -1, 1, 1)
                if res is not None:
                    ref = ref + res
                if relu:
                    ref = ref.relu()
                out = bias_act_(y.clone(memory_format=torch.preserve_format), b, res, relu)
                assert torch.equal(out, ref)
        # non-finite activations must surface, exactly as through F.relu
        k = torch.arange(y.numel(), device='cuda').view(y.shape)
        bad = torch.where(k % 7 == 0, torch.full_like(y, float('nan')), y)
        bad = torch.where(k % 11 == 1, torch.full_like(y, float('inf')), bad)
        bad = torch.where(k % 13 == 2, torch.full_like(y, float('-inf')), bad).contiguous(memory_format=fmt)
        want_bad = (bad + b.view(1, -1, 1, 1)).relu()
        got_bad = bias_act_(bad.clone(memory_format=torch.preserve_format), b, None, True)
        assert torch.equal(torch.isnan(got_bad), torch.isnan(want_bad)) and bool(torch.isnan(got_bad).any())
        assert torch.equal(torch.nan_to_num(got_bad, nan=-1.0), torch.nan_to_num(want_bad, nan=-1.0))


@pytest.mark.gpu
def test_dense_part_captures_into_a_hip_graph():
    """Network + decoding (R3Det.dense_test) has static shapes and no host synchronisation: it must
    stay capturable (the custom ops enqueue on the capturing stream, nothing copies from the host),
    and the replayed graph must give the eager result up to the convolutions' own run-to-run noise."""
    from r3det.models import R3Det
    from r3det.models.detectors import GraphedDense, calibrate_score_bias
    torch.manual_seed(11)
    dev = torch.device('cuda')
    m = R3Det().eval().to(dev)
    img = torch.randn(2, 3, 256, 256, device=dev)
    calibrate_score_bias(m, img, frac=0.02)
    boxes, scores = m.dense_test(img)
    g = GraphedDense(m, img)
    for trial in range(2):
        gb, gs = g(img if trial == 0 else img.clone())
        assert gb.shape == boxes.shape and gs.shape == scores.shape
        assert torch.allclose(gs, scores, atol=1e-4) and torch.allclose(gb, boxes, rtol=1e-3, atol=1e-2)
    res = g.simple_test(img)
    assert len(res) == 2 and all(d.size(1) == 6 and d.size(0) > 0 for d, _ in res)


@pytest.mark.gpu
def test_rretinanet_full_size_config1():
    """BASELINE configs[1]: rretinanet_obb_r50_fpn v1, batch 2 x 1024^2.  The head's nms_pre top-2000 per level
    gives 8576-box pools (2000 + 2000 + 2000 + 2000 + 576); the batched multiclass NMS must return, image by
    image, exactly what the per-image operator path returns (counts, order, labels), and simple_test the same."""
    from r3det.core.post_processing import multiclass_nms_rotated, multiclass_nms_rotated_batch
    from r3det.models import RRetinaNet
    from r3det.models.detectors import calibrate_score_bias
    torch.manual_seed(13)
    dev = torch.device('cuda')
    m = RRetinaNet().eval().to(dev)
    img = torch.randn(2, 3, 1024, 1024, device=dev)
    calibrate_score_bias(m, img, frac=0.01)
    with torch.no_grad():
        cls, reg = m.bbox_head(m.extract_feat(img))
        assert [tuple(c.shape[-2:]) for c in cls] == [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]
        boxes, scores = m.bbox_head.decode_bboxes(cls, reg, img.shape[-2:], m.test_cfg)
        assert boxes.shape == (2, 8576, 5) and scores.shape == (2, 8576, 16)
        cfg = m.test_cfg
        got = multiclass_nms_rotated_batch(boxes, scores, cfg['score_thr'], cfg['nms'], cfg['max_per_img'])
        for i, (d, l) in enumerate(got):
            wd, wl = multiclass_nms_rotated(boxes[i], scores[i], cfg['score_thr'], cfg['nms'], cfg['max_per_img'])
            assert d.size(0) > 50 and torch.equal(d, wd) and torch.equal(l, wl), i
        res = m.simple_test(img)
    assert len(res) == 2 and all(r[0].size(1) == 6 and 0 < r[0].size(0) <= 2000 for r in res)
    # (MIOpen may pick another kernel for the repeated forward: counts agree to the rounding of a few scores)
    assert all(abs(r[0].size(0) - g[0].size(0)) <= max(5, g[0].size(0) // 50) for r, g in zip(res, got))


@pytest.mark.gpu
def test_r3det_full_size_config2_composed_as_the_bench_composes_it():
    """BASELINE configs[2]: r3det_r50_fpn v1, batch 4 x 1024^2, built exactly as bench.py builds it (conv + BN fused,
    channels_last).  Stage by stage on the full-size tensors: the FeatureRefineModule's two-launch channels_last levels
    call gives, bit for bit, x + (P + sample(P)) with P = (conv_5_1(conv_1_5(x)) + bias) + (conv_1_1(x) + bias) put
    together from torch ops and the plain sampler operator, and the module's own output is that up to the convolutions'
    run-to-run noise; the refine head's pools are 5344 boxes per
    image; the batched multiclass NMS returns, image by image, exactly what the per-image operator path returns; and
    the bench's own step (simple_test + pack + gather) reports the same counts."""
    import os
    import sys
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import bench
    from r3det.core.post_processing import multiclass_nms_rotated, multiclass_nms_rotated_batch
    from r3det.ops.feature_refine import fr_forward_nhwc, fr_module_levels_nhwc
    dev = torch.device('cuda')
    model, img = bench.build_model(dev, 21, "R3Det", batch=4)
    assert img.shape == (4, 3, 1024, 1024)
    with torch.no_grad():
        x = model.extract_feat(img)
        assert [tuple(f.shape) for f in x] == [(4, 256, s, s) for s in (128, 64, 32, 16, 8)]
        cls, reg = model.bbox_head(x)
        rois = model.bbox_head.filter_bboxes(cls, reg)
        frm = model.feat_refine_module[0]
        x_refine = frm(x, rois)
        per_level = [torch.cat(lvl).contiguous() for lvl in zip(*rois)]
        # (MIOpen's convolutions of the coarse levels are not bit-reproducible run to run -- 1e-7 of the scale -- so the
        # module's two convolutions run ONCE here and feed both the library call the module makes and the composition)
        ras = [F.conv2d(frm.conv_1_5(f), frm.conv_5_1.weight, None, frm.conv_5_1.stride, frm.conv_5_1.padding) for f in x]
        rbs = [F.conv2d(f, frm.conv_1_1.weight, None, frm.conv_1_1.stride, frm.conv_1_1.padding) for f in x]
        fused = [torch.full_like(f, float('nan')) for f in x]
        assert fr_module_levels_nhwc(ras, rbs, frm.conv_5_1.bias, frm.conv_1_1.bias, x, per_level,
                                     [fr.spatial_scale for fr in frm.fr], frm.fr[0].points, fused)
        for lvl, (f, ra, rb, got, mod, boxes, fr) in enumerate(zip(x, ras, rbs, fused, x_refine, per_level, frm.fr)):
            assert f.is_contiguous(memory_format=torch.channels_last) and mod.is_contiguous(memory_format=torch.channels_last)
            P = (ra + frm.conv_5_1.bias.view(1, -1, 1, 1)) + (rb + frm.conv_1_1.bias.view(1, -1, 1, 1))
            P = P.contiguous(memory_format=torch.channels_last)
            o = torch.empty_like(P)
            assert fr_forward_nhwc(P, boxes, fr.spatial_scale, fr.points, o)
            assert torch.equal(got, f + o), lvl
            assert (mod - got).abs().max().item() <= 1e-6 * got.abs().max().item(), lvl  # (the module's own call)
        cls, reg = model.refine_head[0](x_refine)
        boxes, scores = model.refine_head[-1].decode_bboxes(cls, reg, img.shape[-2:], model.test_cfg, rois=rois)
        assert boxes.shape == (4, 5344, 5) and scores.shape == (4, 5344, 16)
        cfg = model.test_cfg
        got = multiclass_nms_rotated_batch(boxes, scores, cfg['score_thr'], cfg['nms'], cfg['max_per_img'])
        for i, (d, l) in enumerate(got):
            wd, wl = multiclass_nms_rotated(boxes[i], scores[i], cfg['score_thr'], cfg['nms'], cfg['max_per_img'])
            assert d.size(0) > 50 and torch.equal(d, wd) and torch.equal(l, wl), i
        counts = bench.model_step(model, img, batch_size=4)
    # (MIOpen may pick another kernel for the repeated forward: counts agree to the rounding of a few scores)
    assert all(abs(int(c) - g[0].size(0)) <= max(5, g[0].size(0) // 50) for c, g in zip(counts.tolist(), got))


@pytest.mark.gpu
def test_whole_step_in_one_hip_graph_without_host_synchronisation():
    """Round 5: network + decoding + pool + multiclass NMS + the padded result as ONE graph; ten
    steady-state steps under torch's sync debug mode (any implicit host synchronisation raises); the result equals
    the list form of the same dense outputs."""
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    from r3det.models import R3Det
    from r3det.models.detectors import GraphedStep, calibrate_score_bias
    torch.manual_seed(11)
    dev = torch.device('cuda')
    m = R3Det().eval().to(dev)
    img = torch.randn(2, 3, 256, 256, device=dev)
    calibrate_score_bias(m, img, frac=0.02)
    g = GraphedStep(m, img)
    g.step(img)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for _ in range(10):
            out, redo = g.step(g.static_in)
            assert not redo
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    cfg = m.test_cfg
    rows = cfg['max_per_img']
    assert out.shape == (2, rows + 1, 7)
    # the dense outputs the graph produced, through the list form
    boxes, scores = m.dense_test(img)
    lists = multiclass_nms_rotated_batch(boxes, scores, cfg['score_thr'], cfg['nms'], rows)
    res = g.nms.lists()
    for (d, lab), (d2, lab2) in zip(lists, res):
        assert d.size(0) > 0 and abs(d.size(0) - d2.size(0)) <= max(2, d.size(0) // 50)  # (conv run-to-run noise moves a few scores across the threshold)
    # ... and exactly, on the graph's own pool: replay, then the list form on what dense_test left in the graph's buffers
    o = out.clone()
    for i, (d2, lab2) in enumerate(res):
        k = d2.size(0)
        assert int(o[i, rows, 0]) == k and not bool(o[i, k:rows].any())
        assert torch.equal(o[i, :k, :6], d2) and torch.equal(o[i, :k, 6].long(), lab2)


@pytest.mark.gpu
@pytest.mark.parametrize("lag", [1, 2])
def test_graphed_step_recovers_from_a_short_capacity(lag):
    from r3det.models import R3Det
    from r3det.models.detectors import GraphedStep, calibrate_score_bias
    torch.manual_seed(12)
    dev = torch.device('cuda')
    m = R3Det().eval().to(dev)
    img = torch.randn(1, 3, 256, 256, device=dev)
    calibrate_score_bias(m, img, frac=0.05)
    g = GraphedStep(m, img, cap=64, lag=lag)       # far too small
    out, redo = g.step(img)
    assert not redo
    short = int(out[0, -1, 0])
    redone = 0
    for _ in range(16):                   # the next steps report it, grow the capacity and record again
        out, redo = g.step(img)
        assert redo in (0, lag)           # `redo` = how many of the most recent steps to run again
        redone += redo
        if not redo and redone and not g.flush():
            break
    assert redone and g.nms.cap > 64
    assert g.flush() == 0 and g.nms.pending() == 0
    assert int(out[0, -1, 0]) >= short and int(g.nms.overflow[0]) == 0


@pytest.mark.gpu
def test_graphed_step_flush_reports_the_last_steps_of_a_loop():
    """the last step of a step() loop was never checked.  flush() looks at every step still pending."""
    from r3det.models import R3Det
    from r3det.models.detectors import GraphedStep, calibrate_score_bias
    torch.manual_seed(12)
    dev = torch.device('cuda')
    m = R3Det().eval().to(dev)
    img = torch.randn(1, 3, 256, 256, device=dev)
    calibrate_score_bias(m, img, frac=0.05)
    g = GraphedStep(m, img, cap=64, lag=2)
    out, redo = g.step(img)
    assert redo == 0 and g.nms.pending() == 1
    assert g.flush() == 1 and g.nms.cap > 64 and g.nms.pending() == 0     # that one step has to be run again
    while True:                                                              # ... until the capacity holds the pool
        g.step(img)
        if not g.flush():
            break
    assert int(g.nms.overflow[0]) == 0


@pytest.mark.gpu
def test_graphed_simple_test_with_a_short_capacity_equals_the_list_form():
    """simple_test must not return the result of a pool that outgrew the capacity -- the lists equal
    multiclass_nms_rotated's (models/detectors/r3det.py:112-143, core/post_processing/bbox_nms_rotated.py:7-77) on the
    same dense outputs although the graph was recorded with room for 64 candidates."""
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    from r3det.models import R3Det
    from r3det.models.detectors import GraphedStep, calibrate_score_bias
    torch.manual_seed(14)
    dev = torch.device('cuda')
    m = R3Det().eval().to(dev)
    img = torch.randn(2, 3, 256, 256, device=dev)
    calibrate_score_bias(m, img, frac=0.05)
    g = GraphedStep(m, img, cap=64)
    res = g.simple_test(img)
    assert g.nms.cap > 64 and g.nms.overflow.tolist() == [0, 0] and g.nms.pending() == 0
    cfg = m.test_cfg
    boxes, scores = m.dense_test(img)
    n_cand = (scores[..., :-1] > cfg['score_thr']).flatten(1).sum(1)
    assert int(n_cand.min()) > 64          # every image's pool is beyond the capacity the graph was first recorded with
    lists = multiclass_nms_rotated_batch(boxes, scores, cfg['score_thr'], cfg['nms'], cfg['max_per_img'])
    for (d, lab), (d2, lab2) in zip(lists, res):
        # (the convs are not run-to-run bit-stable: a few scores cross the threshold between two passes)
        assert d.size(0) > 0 and abs(d.size(0) - d2.size(0)) <= max(2, d.size(0) // 50)
        k = min(d.size(0), d2.size(0), 20)
        assert torch.allclose(d[:k, 5], d2[:k, 5], atol=1e-4)      # the same top detections, in the same order
    # a second call on the grown graph: no further growth, the same lists
    cap = g.nms.cap
    res2 = g.simple_test(img)
    assert g.nms.cap == cap
    for r, r2 in zip(res, res2):
        assert abs(r[0].size(0) - r2[0].size(0)) <= max(2, r[0].size(0) // 50)


@pytest.mark.gpu
def test_simple_test_with_img_metas_returns_what_the_reference_returns():
    """``simple_test(img, img_metas, rescale)`` (models/detectors/r3det.py:112-143): per image a list over the classes of
    (k, 6) float32 ndarrays (rbbox2result); rescale divides cx, cy, w, h by scale_factor BEFORE the NMS
    (rotate_anchor_head.py:657-660), the angle stays."""
    import numpy as np

    from r3det.core.post_processing import multiclass_nms_rotated
    from r3det.models import R3Det
    from r3det.models.detectors import calibrate_score_bias
    torch.manual_seed(13)
    dev = torch.device('cuda')
    m = R3Det().eval().to(dev)
    img = torch.randn(2, 3, 256, 256, device=dev)
    calibrate_score_bias(m, img, frac=0.02, per_class=True)
    metas = [dict(img_shape=(256, 256, 3), scale_factor=np.array([2.0, 2.0, 2.0, 2.0], np.float32)),
             dict(img_shape=(256, 256, 3), scale_factor=0.5)]
    res = m.simple_test(img, metas)          # the structure the reference returns
    assert len(res) == 2 and all(len(r) == 15 for r in res)
    assert all(a.dtype == np.float32 and a.shape[1] == 6 for r in res for a in r) and sum(a.shape[0] for a in res[0]) > 0
    # the values, on ONE set of dense outputs (two passes of the network differ in their last bits: MIOpen's atomics)
    from r3det.models.detectors import _test_results
    boxes, scores = m.dense_test(img)
    cfg, head = m.test_cfg, m.refine_head[-1]
    for rescale in (False, True):
        got = _test_results(head, boxes, scores, cfg, metas, rescale, None)
        for i, sf in enumerate((2.0, 0.5)):
            b = boxes[i].clone()
            if rescale:
                b[:, :4] = b[:, :4] / sf     # (before the NMS; the angle stays: rotate_anchor_head.py:657-660)
            d, lab = multiclass_nms_rotated(b, scores[i], cfg['score_thr'], cfg['nms'], cfg['max_per_img'])
            assert d.size(0) > 0 and sum(a.shape[0] for a in got[i]) == d.size(0)
            for c, a in enumerate(got[i]):
                assert np.array_equal(a, d[lab == c].cpu().numpy())
