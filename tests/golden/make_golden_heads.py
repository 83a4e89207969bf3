"""Golden vectors for the callers either side of the FR sampler and for the training glue: run the
REFERENCE's own Python --

  r3det/core/bbox/coder/delta_xywha_rbbox_coder.py     delta2bbox_v1 / bbox2delta_v1 / the coder class
  r3det/core/bbox/rtransforms.py                        obb2hbb, obb2poly, obb2xyxy, hbb2obb, poly2obb ...
  r3det/core/anchor/{ranchor_generator,rutils}.py       RAnchorGenerator (on a stand-in base), inside flags
  r3det/models/dense_heads/rotate_retina_head.py        RRetinaHead.filter_bboxes
  r3det/models/dense_heads/rotate_retina_refine_head.py RRetinaRefineHead.refine_bboxes / .loss
  r3det/models/dense_heads/rotate_anchor_head.py        RAnchorHead.loss / get_targets / _get_targets_single
  r3det/core/bbox/iou_calculators/rotate_iou2d_calculator.py   RBboxOverlaps2D_v1

-- from where it lies, on CPU tensors in the build container, and record inputs + outputs
(tests/golden/heads.npz: data only).

mmcv / mmdet are not installed and not under /root/reference.  The reference files above import
names from them; those names are bound here to STAND-INS (marked "mmdet stand-in" below, restated
from memory of mmdet 2.19): AnchorGenerator, PseudoSampler, images_to_levels / multi_apply / unmap,
and -- taken from this repo's own restatements -- MaxIoUAssigner, FocalLoss, SmoothL1Loss.  So the
recorded losses pin the reference's OWN glue (targets, masks, normalisation, level split) and its
coder, not the third-party pieces.  ``rbbox_iou`` inside the reference's IoU calculator is bound to
oracle/_ref (the reference's own CPU code of the same arithmetic).

    python tests/golden/make_golden_heads.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
from oracle import api as O  # noqa: E402

REF = os.environ.get("R3DET_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _mod(name, is_pkg=False, **attrs):
    m = types.ModuleType(name)
    if is_pkg:
        m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class _Registry:
    def __init__(self):
        self.modules = {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.modules[cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg, **kw):
        cfg = dict(cfg)
        return self.modules[cfg.pop('type')](**cfg, **kw)


class Cfg(dict):
    """mmcv.Config-like: attribute access, nested."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return Cfg(v) if isinstance(v, dict) else v


# ----------------------------------------------------------------------------- mmdet stand-ins
class AnchorGenerator:
    """mmdet stand-in (mmdet/core/anchor/anchor_generator.py, 2.19), the parts the rotated heads use."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.):
        from torch.nn.modules.utils import _pair
        self.strides = [_pair(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else base_sizes
        if scales is not None:
            self.scales = torch.Tensor(scales)
        else:
            octave_scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
            self.scales = torch.Tensor(octave_scales * octave_base_scale)
        self.ratios = torch.Tensor(ratios)
        self.scale_major, self.center_offset = scale_major, center_offset
        self.base_anchors = [self.gen_single_level_base_anchors(b, self.scales, self.ratios) for b in self.base_sizes]

    @property
    def num_levels(self):
        return len(self.strides)

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    def gen_single_level_base_anchors(self, base_size, scales, ratios):
        w = h = base_size
        x_center, y_center = self.center_offset * w, self.center_offset * h
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
        hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
        return torch.stack([x_center - 0.5 * ws, y_center - 0.5 * hs, x_center + 0.5 * ws, y_center + 0.5 * hs], dim=-1)

    def single_level_grid_priors(self, featmap_size, level_idx, dtype=torch.float32, device='cuda'):
        base = self.base_anchors[level_idx].to(device).to(dtype)
        feat_h, feat_w = featmap_size
        stride_w, stride_h = self.strides[level_idx]
        shift_x = torch.arange(0, feat_w, device=device).to(dtype) * stride_w
        shift_y = torch.arange(0, feat_h, device=device).to(dtype) * stride_h
        xx = shift_x.repeat(len(shift_y))
        yy = shift_y.view(-1, 1).repeat(1, len(shift_x)).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1)
        return (base[None, :, :] + shifts[:, None, :]).view(-1, 4)

    def grid_priors(self, featmap_sizes, device='cuda'):
        return [self.single_level_grid_priors(featmap_sizes[i], i, device=device) for i in range(self.num_levels)]

    def valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        out = []
        for i in range(self.num_levels):
            feat_h, feat_w = featmap_sizes[i]
            h, w = pad_shape[:2]
            valid_h = min(int(np.ceil(h / self.strides[i][1])), feat_h)
            valid_w = min(int(np.ceil(w / self.strides[i][0])), feat_w)
            vx = torch.zeros(feat_w, dtype=torch.bool, device=device)
            vy = torch.zeros(feat_h, dtype=torch.bool, device=device)
            vx[:valid_w] = 1
            vy[:valid_h] = 1
            xx = vx.repeat(len(vy))
            yy = vy.view(-1, 1).repeat(1, len(vx)).view(-1)
            valid = xx & yy
            A = self.num_base_anchors[i]
            out.append(valid[:, None].expand(valid.size(0), A).contiguous().view(-1))
        return out


def multi_apply(func, *args, **kwargs):
    from functools import partial
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def images_to_levels(target, num_levels):
    target = torch.stack(target, 0)
    out, start = [], 0
    for n in num_levels:
        out.append(target[:, start:start + n])
        start += n
    return out


def unmap(data, count, inds, fill=0):
    if data.dim() == 1:
        ret = data.new_full((count, ), fill)
        ret[inds.type(torch.bool)] = data
    else:
        ret = data.new_full((count, ) + data.size()[1:], fill)
        ret[inds.type(torch.bool), :] = data
    return ret


class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :] if gt_bboxes.numel() else gt_bboxes.view(-1, 5)


class PseudoSampler:
    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return SamplingResult(pos, neg, bboxes, gt_bboxes, assign_result)


class ConvModule(nn.Sequential):
    def __init__(self, cin, cout, k, stride=1, padding=0, conv_cfg=None, norm_cfg=None, **kw):
        super().__init__(nn.Conv2d(cin, cout, k, stride=stride, padding=padding), nn.ReLU())


def bind_reference():
    """Import the reference modules with the third-party names bound to the stand-ins above."""
    # third-party pieces taken from this repo's restatements (see the module docstring): imported first,
    # then this repo's ``r3det`` is dropped from sys.modules so that the name can carry the reference's files
    from r3det.core.bbox.assigners import MaxIoUAssigner
    from r3det.models.losses import build_loss
    for k in [k for k in sys.modules if k == 'r3det' or k.startswith('r3det.')]:
        del sys.modules[k]
    ident = lambda *a, **k: (lambda f: f)  # noqa: E731
    HEADS, CODERS, IOUS, ANCHORS = _Registry(), _Registry(), _Registry(), _Registry()
    _mod("cv2")
    _mod("mmcv", True, jit=ident)
    _mod("mmcv.cnn", ConvModule=ConvModule, bias_init_with_prob=lambda p: float(-np.log((1 - p) / p)),
         normal_init=lambda m, std=0.01, bias=0: None)
    _mod("mmcv.runner", force_fp32=ident)
    _mod("mmcv.ops", box_iou_rotated=None, nms_rotated=None)
    _mod("mmdet", True)
    _mod("mmdet.core", True)
    _mod("mmdet.core.bbox", True)
    _mod("mmdet.core.bbox.builder", BBOX_CODERS=CODERS)
    _mod("mmdet.core.bbox.coder", True)
    _mod("mmdet.core.bbox.coder.base_bbox_coder", BaseBBoxCoder=type("BaseBBoxCoder", (), {}))
    _mod("mmdet.core.bbox.iou_calculators", True)
    _mod("mmdet.core.bbox.iou_calculators.builder", IOU_CALCULATORS=IOUS)
    _mod("mmdet.core.anchor", True, AnchorGenerator=AnchorGenerator)
    _mod("mmdet.core.anchor.builder", ANCHOR_GENERATORS=ANCHORS)

    def rbbox_iou(b1, b2, vec=False, iof=False):
        assert not vec
        return torch.from_numpy(O.ref_v1_iou_mat(b1.contiguous().numpy(), b2.contiguous().numpy(), iof=iof))

    _mod("r3det", True)
    _mod("r3det.ops", True, rbbox_iou=rbbox_iou, obb_overlaps=None, convex_sort=None, batched_rnms=None,
         ml_nms_rotated=None, obb_batched_nms=None)
    _mod("r3det.core", True)
    _mod("r3det.core.bbox", True)
    _mod("r3det.core.bbox.coder", True)
    coder = _load("r3det.core.bbox.coder.delta_xywha_rbbox_coder", "r3det/core/bbox/coder/delta_xywha_rbbox_coder.py")
    rt = _load("r3det.core.bbox.rtransforms", "r3det/core/bbox/rtransforms.py")
    _mod("r3det.core.bbox.iou_calculators", True)
    calc = _load("r3det.core.bbox.iou_calculators.rotate_iou2d_calculator",
                 "r3det/core/bbox/iou_calculators/rotate_iou2d_calculator.py")
    _mod("r3det.core.anchor", True)
    ag = _load("r3det.core.anchor.ranchor_generator", "r3det/core/anchor/ranchor_generator.py")
    ru = _load("r3det.core.anchor.rutils", "r3det/core/anchor/rutils.py")
    core = sys.modules["r3det.core"]
    core.multiclass_nms_rotated, core.obb2hbb, core.ranchor_inside_flags = None, rt.obb2hbb, ru.ranchor_inside_flags

    def build_assigner(cfg):
        cfg = dict(cfg)
        assert cfg.pop('type') == 'MaxIoUAssigner'
        cfg.pop('iou_calculator')
        a = MaxIoUAssigner(**cfg)
        a.iou_calculator = calc.RBboxOverlaps2D_v1()  # the reference's calculator class
        return a

    md = sys.modules["mmdet.core"]
    md.build_assigner = build_assigner
    md.build_bbox_coder = lambda cfg: CODERS.build(cfg)
    md.build_prior_generator = lambda cfg: ANCHORS.build(cfg)
    md.build_sampler = lambda cfg, **kw: PseudoSampler()
    md.images_to_levels, md.multi_apply, md.unmap = images_to_levels, multi_apply, unmap
    _mod("mmdet.models", True)
    _mod("mmdet.models.builder", HEADS=HEADS, build_loss=lambda cfg: build_loss(dict(cfg)))
    _mod("mmdet.models.dense_heads", True)
    _mod("mmdet.models.dense_heads.base_dense_head", BaseDenseHead=nn.Module)
    _mod("r3det.models", True)
    _mod("r3det.models.dense_heads", True)
    ah = _load("r3det.models.dense_heads.rotate_anchor_head", "r3det/models/dense_heads/rotate_anchor_head.py")
    sys.modules["r3det.models.dense_heads"].RAnchorHead = ah.RAnchorHead
    rh = _load("r3det.models.dense_heads.rotate_retina_head", "r3det/models/dense_heads/rotate_retina_head.py")
    sys.modules["r3det.models.dense_heads"].RRetinaHead = rh.RRetinaHead
    rr = _load("r3det.models.dense_heads.rotate_retina_refine_head",
               "r3det/models/dense_heads/rotate_retina_refine_head.py")
    return types.SimpleNamespace(coder=coder, rt=rt, calc=calc, ag=ag, ru=ru, RRetinaHead=rh.RRetinaHead,
                                 RRetinaRefineHead=rr.RRetinaRefineHead)


LOSS_CLS = dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0)
LOSS_BBOX = dict(type='SmoothL1Loss', beta=0.11, loss_weight=1.0)
ANCHOR_CFG = dict(type='RAnchorGenerator', octave_base_scale=4, scales_per_octave=3, ratios=[1.0, 0.5, 2.0],
                  strides=[8, 16, 32, 64, 128])
CODER_CFG = dict(type='DeltaXYWHAOBBoxCoder', target_means=(.0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0))


def train_cfg(pos, neg):
    return Cfg(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=pos, neg_iou_thr=neg, min_pos_iou=0, ignore_iof_thr=-1,
                             iou_calculator=dict(type='RBboxOverlaps2D_v1')),
               allowed_border=-1, pos_weight=-1, debug=False)


def gt_boxes(n, seed, size):
    r = np.random.default_rng(seed)
    w = np.exp(r.uniform(np.log(size / 16), np.log(size / 2), n))
    h = np.maximum(w / np.exp(r.uniform(0, np.log(4), n)), 4)
    return np.stack([r.uniform(0, size, n), r.uniform(0, size, n), w, h, r.uniform(-np.pi / 2, 0, n)], 1).astype(np.float32)


def main():
    assert os.path.isdir(REF)
    R = bind_reference()
    out = {}
    g = torch.Generator().manual_seed(7)

    # ---- coder: direct cases (clipped dw / dh, max_shape clamp, multi-class deltas, non-unit stds)
    rois = torch.from_numpy(gt_boxes(300, 1, 512))
    deltas = torch.randn(300, 5, generator=g) * torch.tensor([0.5, 0.5, 3.0, 3.0, 0.7])  # |dw| up to ~9 > 4.135
    out["coder_rois"], out["coder_deltas"] = rois.numpy(), deltas.numpy()
    out["coder_decode"] = R.coder.delta2bbox_v1(rois, deltas).numpy()
    out["coder_decode_clamped"] = R.coder.delta2bbox_v1(rois, deltas, max_shape=(512, 384)).numpy()
    out["coder_decode_stds"] = R.coder.delta2bbox_v1(rois, deltas, (0.1, -0.1, 0.2, 0., 0.05),
                                                     (0.5, 0.5, 0.25, 0.25, 0.1)).numpy()
    d15 = torch.randn(300, 15, generator=g)
    out["coder_deltas15"], out["coder_decode15"] = d15.numpy(), R.coder.delta2bbox_v1(rois, d15).numpy()
    gts = torch.from_numpy(gt_boxes(300, 2, 512))
    out["coder_gts"] = gts.numpy()
    out["coder_encode"] = R.coder.bbox2delta_v1(rois, gts).numpy()
    out["coder_encode_stds"] = R.coder.bbox2delta_v1(rois, gts, (0.1, -0.1, 0.2, 0., 0.05),
                                                     (0.5, 0.5, 0.25, 0.25, 0.1)).numpy()
    c = R.coder.DeltaXYWHAOBBoxCoder()
    out["coder_class_roundtrip"] = c.decode(rois, c.encode(rois, gts)).numpy()

    # ---- rtransforms
    b = torch.from_numpy(gt_boxes(200, 3, 800))
    b3 = b.clone()
    b3[:, 4] = torch.rand(200, generator=g) * np.pi - np.pi / 2
    b2 = b.clone()
    b2[:, 4] = torch.rand(200, generator=g) * np.pi - np.pi / 4
    out["rt_obb_v1"], out["rt_obb_v2"], out["rt_obb_v3"] = b.numpy(), b2.numpy(), b3.numpy()
    for v, x in (("v1", b), ("v2", b2), ("v3", b3)):
        out[f"rt_obb2hbb_{v}"] = R.rt.obb2hbb(x, v).numpy()
        out[f"rt_obb2poly_{v}"] = R.rt.obb2poly(x, v).numpy()
        out[f"rt_obb2xyxy_{v}"] = R.rt.obb2xyxy(x, v).numpy()
        out[f"rt_poly2obb_{v}"] = R.rt.poly2obb(R.rt.obb2poly(x, v), v).numpy()
        x6 = np.hstack([x.numpy(), np.linspace(0, 1, 200, dtype=np.float32)[:, None]])
        out[f"rt_obb2poly_np_{v}"] = np.asarray(R.rt.obb2poly_np(x6, v))
        hb = R.rt.obb2xyxy(x, v)
        out[f"rt_hbb2obb_{v}"] = R.rt.hbb2obb(hb, v).numpy()
        out[f"rt_norm_angle_{v}"] = np.asarray(R.rt.norm_angle(np.linspace(-7, 7, 57), v))
    # (poly2obb_np is not recorded: v1 / v3 need cv2, v2 calls np.float, which numpy 2 no longer has)
    out["rt_rbbox2roi"] = R.rt.rbbox2roi([b[:7], b[:0], b[7:19]]).numpy()
    lab = torch.randint(0, 15, (200,), generator=g)
    d6 = torch.cat([b, torch.rand(200, 1, generator=g)], 1)
    res = R.rt.rbbox2result(d6, lab, 15)
    out["rt_result_dets"], out["rt_result_labels"] = d6.numpy(), lab.numpy()
    out["rt_result_sizes"] = np.array([len(r) for r in res])
    out["rt_result_cat"] = np.concatenate(res)

    # ---- anchors (reference subclass on the stand-in base) and inside flags
    gen = R.ag.RAnchorGenerator(strides=[8, 16, 32, 64, 128], ratios=[1.0, 0.5, 2.0], octave_base_scale=4,
                                scales_per_octave=3)
    sizes = [(16, 12), (8, 6), (4, 3), (2, 2), (1, 1)]
    out["anchor_sizes"] = np.array(sizes)
    for i, a in enumerate(gen.grid_priors(sizes, device='cpu')):
        out[f"anchors_l{i}"] = a.numpy()
    out["anchors_1024_l0_head"] = gen.single_level_grid_priors((128, 128), 0, device='cpu')[:2000].numpy()
    for i, f in enumerate(gen.valid_flags(sizes, (100, 90, 3), device='cpu')):
        out[f"valid_flags_l{i}"] = f.numpy()
    fa = torch.cat(gen.grid_priors(sizes, device='cpu'))
    vf = torch.cat(gen.valid_flags(sizes, (100, 90, 3), device='cpu'))
    out["inside_b0"] = R.ru.ranchor_inside_flags(fa, vf, (100, 90), 0).numpy()
    out["inside_b16"] = R.ru.ranchor_inside_flags(fa, vf, (100, 90), 16).numpy()
    out["inside_bneg"] = R.ru.ranchor_inside_flags(fa, vf, (100, 90), -1).numpy()

    # ---- heads: producers of the FR boxes + training losses, two image sizes
    for tag, (H, W), n_gt in (("a", (128, 96), 12), ("b", (64, 64), 5)):
        N, C, A = 2, 15, 9
        sizes = [(max(1, -(-H // s)), max(1, -(-W // s))) for s in (8, 16, 32, 64, 128)]
        head = R.RRetinaHead(num_classes=C, in_channels=8, stacked_convs=1, feat_channels=8,
                             anchor_generator=ANCHOR_CFG, bbox_coder=CODER_CFG, loss_cls=LOSS_CLS, loss_bbox=LOSS_BBOX,
                             train_cfg=train_cfg(0.5, 0.4))
        cls = [torch.randn(N, A * C, h, w, generator=g) * 2 for h, w in sizes]
        reg = [torch.randn(N, A * 5, h, w, generator=g) * torch.tensor([0.3, 0.3, 1.5, 1.5, 0.4]).repeat(A)[None, :, None, None]
               for h, w in sizes]
        # ties between anchors at some positions: argmax must take the first
        cls[0][0, :, 0, 0] = 1.0
        cls[1][1, :C, 1, 1] = cls[1][1, C:2 * C, 1, 1] = 5.0
        gtb = [torch.from_numpy(gt_boxes(n_gt, 10 + i, min(H, W))) for i in range(N)]
        gtl = [torch.randint(0, C, (n_gt,), generator=g) for _ in range(N)]
        if tag == "b":
            gtb[1], gtl[1] = gtb[1][:0], gtl[1][:0]  # an image without ground truth
        metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0) for _ in range(N)]
        out[f"{tag}_sizes"] = np.array(sizes)
        for l in range(5):
            out[f"{tag}_cls_l{l}"], out[f"{tag}_reg_l{l}"] = cls[l].numpy(), reg[l].numpy()
        for i in range(N):
            out[f"{tag}_gt_bboxes_{i}"], out[f"{tag}_gt_labels_{i}"] = gtb[i].numpy(), gtl[i].numpy()
        rois = head.filter_bboxes(cls, reg)
        for i in range(N):
            for l in range(5):
                out[f"{tag}_rois_{i}_l{l}"] = rois[i][l].numpy()
        losses = head.loss(cls, reg, gtb, gtl, metas)
        out[f"{tag}_s0_loss_cls"] = np.array([float(v) for v in losses['loss_cls']], dtype=np.float64)
        out[f"{tag}_s0_loss_bbox"] = np.array([float(v) for v in losses['loss_bbox']], dtype=np.float64)
        anchor_list, valid = head.get_anchors(sizes, metas, device='cpu')
        tg = head.get_targets(anchor_list, valid, gtb, metas, gt_bboxes_ignore_list=None, gt_labels_list=gtl,
                              label_channels=C)
        for l in range(5):
            out[f"{tag}_s0_labels_l{l}"], out[f"{tag}_s0_label_weights_l{l}"] = tg[0][l].numpy(), tg[1][l].numpy()
            out[f"{tag}_s0_bbox_targets_l{l}"], out[f"{tag}_s0_bbox_weights_l{l}"] = tg[2][l].numpy(), tg[3][l].numpy()
        out[f"{tag}_s0_num_total_pos"] = np.array(tg[4])
        # refinement head on those rois
        rhead = R.RRetinaRefineHead(num_classes=C, in_channels=8, stacked_convs=1, feat_channels=8,
                                    assign_by_circumhbbox=None, bbox_coder=CODER_CFG, loss_cls=LOSS_CLS,
                                    loss_bbox=LOSS_BBOX, train_cfg=train_cfg(0.6, 0.5))
        rcls = [torch.randn(N, C, h, w, generator=g) * 2 for h, w in sizes]
        rreg = [torch.randn(N, 5, h, w, generator=g) * torch.tensor([0.2, 0.2, 0.5, 0.5, 0.2])[None, :, None, None]
                for h, w in sizes]
        for l in range(5):
            out[f"{tag}_rcls_l{l}"], out[f"{tag}_rreg_l{l}"] = rcls[l].numpy(), rreg[l].numpy()
        rl = rhead.loss(rcls, rreg, gtb, gtl, metas, rois=rois)
        out[f"{tag}_sr_loss_cls"] = np.array([float(v) for v in rl['loss_cls']], dtype=np.float64)
        out[f"{tag}_sr_loss_bbox"] = np.array([float(v) for v in rl['loss_bbox']], dtype=np.float64)
        refined = rhead.refine_bboxes(rcls, rreg, rois=rois)
        for i in range(N):
            for l in range(5):
                out[f"{tag}_refined_{i}_l{l}"] = refined[i][l].numpy()
        print(tag, "s0", out[f"{tag}_s0_loss_cls"].sum(), out[f"{tag}_s0_loss_bbox"].sum(), "npos", tg[4],
              "sr", out[f"{tag}_sr_loss_cls"].sum(), out[f"{tag}_sr_loss_bbox"].sum())
    np.savez_compressed(os.path.join(OUT, "heads.npz"), **out)
    print("done", len(out), "arrays")


if __name__ == "__main__":
    main()
