"""R3Det / RRetinaNet detectors: inference (models/detectors/r3det.py:112-143, rretinanet.py:23-46)
and the training step (r3det.py:75-110; mmdet's SingleStageDetector.forward_train for RRetinaNet)."""
from collections import OrderedDict

import torch
import torch.nn as nn

from ..core.post_processing import multiclass_nms_rotated_batch
from ..ops import FeatureRefineModule
from .backbone import FPN, ResNet50
from .heads import RRetinaHead, RRetinaRefineHead

from .heads import S0_TRAIN_CFG, SR_TRAIN_CFG  # noqa: E402

TEST_CFG = dict(nms_pre=2000, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
# configs/r3det/r3det_r50_fpn_1x_dota_v1.py:69-98
R3DET_TRAIN_CFG = dict(s0=S0_TRAIN_CFG, sr=[SR_TRAIN_CFG], stage_loss_weights=[1.0])


def default_img_metas(img):
    """Synthetic tiles: no padding, no rescale."""
    h, w = img.shape[-2:]
    return [dict(img_shape=(h, w, 3), pad_shape=(h, w, 3), scale_factor=1.0) for _ in range(img.size(0))]


def parse_losses(losses):
    """mmdet BaseDetector._parse_losses: every entry whose key contains 'loss' is summed into the
    total; lists are summed over their (per-level) items.  Returns (loss, log_vars of 0-dim tensors --
    the reference all-reduces and ``.item()``s them for logging, which is not part of the step)."""
    log_vars = OrderedDict()
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, (list, tuple)):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError(f'{name} is not a tensor or list of tensors')
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, log_vars


class _Detector(nn.Module):
    def forward(self, img, img_metas=None, return_loss=True, **kwargs):
        """mmdet BaseDetector.forward: the entry DistributedDataParallel wraps."""
        if return_loss:
            return self.forward_train(img, img_metas or default_img_metas(img), **kwargs)
        return self.simple_test(img)

    def train_step(self, data, optimizer=None):
        """mmdet BaseDetector.train_step: losses -> (loss, log_vars, num_samples)."""
        loss, log_vars = parse_losses(self(**data))
        return dict(loss=loss, log_vars=log_vars, num_samples=data['img'].size(0))


class RRetinaNet(_Detector):
    """configs/rretinanet/rretinanet_obb_r50_fpn_1x_dota_v1.py: one RRetinaHead, assignment on the
    oriented GT (assign_by_circumhbbox=None)."""

    def __init__(self, num_classes=15, test_cfg=None, train_cfg=None):
        super().__init__()
        self.test_cfg = dict(test_cfg or TEST_CFG)
        self.train_cfg = dict(train_cfg or S0_TRAIN_CFG)
        self.backbone = ResNet50()
        self.neck = FPN()
        self.bbox_head = RRetinaHead(num_classes, test_cfg=self.test_cfg, train_cfg=self.train_cfg,
                                     assign_by_circumhbbox=None)

    def extract_feat(self, img):
        return self.neck(self.backbone(img))

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        x = self.extract_feat(img)
        cls, reg = self.bbox_head(x)
        return self.bbox_head.loss(cls, reg, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)

    @torch.no_grad()
    def simple_test(self, img):
        x = self.extract_feat(img)
        cls, reg = self.bbox_head(x)
        return self.bbox_head.get_bboxes(cls, reg, img.shape[-2:], self.test_cfg)


class R3Det(_Detector):
    """num_refine_stages x (FeatureRefineModule -> RRetinaRefineHead) after the base head;
    state-dict names follow the reference (backbone / neck / bbox_head / feat_refine_module.i /
    refine_head.i)."""

    def __init__(self, num_classes=15, num_refine_stages=1, frm_cfgs=None, test_cfg=None, train_cfg=None):
        super().__init__()
        self.test_cfg = dict(test_cfg or TEST_CFG)
        self.train_cfg = dict(train_cfg or R3DET_TRAIN_CFG)
        if len(self.train_cfg['sr']) < num_refine_stages:
            self.train_cfg['sr'] = list(self.train_cfg['sr']) * num_refine_stages
            self.train_cfg['stage_loss_weights'] = list(self.train_cfg['stage_loss_weights']) * num_refine_stages
        self.num_refine_stages = num_refine_stages
        self.backbone = ResNet50()
        self.neck = FPN()
        # the base head assigns on the circumscribed horizontal box of the GT (its default 'v1',
        # rotate_anchor_head.py:47,220-224); the refine heads on the oriented GT (config :58)
        self.bbox_head = RRetinaHead(num_classes, test_cfg=self.test_cfg, train_cfg=self.train_cfg['s0'])
        frm_cfgs = frm_cfgs or [dict(in_channels=256, featmap_strides=[8, 16, 32, 64, 128])] * num_refine_stages
        self.feat_refine_module = nn.ModuleList(FeatureRefineModule(**c) for c in frm_cfgs)
        self.refine_head = nn.ModuleList(RRetinaRefineHead(num_classes, test_cfg=self.test_cfg,
                                                           train_cfg=self.train_cfg['sr'][i])
                                         for i in range(num_refine_stages))
        for m in self.feat_refine_module:
            m.init_weights()

    def extract_feat(self, img):
        return self.neck(self.backbone(img))

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        """r3det.py:75-110: base-head loss ('s0.*'), then per refinement stage the FR module on the
        detached best boxes, the refine head and its loss weighted by ``stage_loss_weights`` ('sr{i}.*')."""
        losses = dict()
        x = self.extract_feat(img)
        outs = self.bbox_head(x)
        for name, value in self.bbox_head.loss(*outs, gt_bboxes, gt_labels, img_metas,
                                               gt_bboxes_ignore=gt_bboxes_ignore).items():
            losses[f's0.{name}'] = value
        rois = self.bbox_head.filter_bboxes(*outs)
        for i in range(self.num_refine_stages):
            lw = self.train_cfg['stage_loss_weights'][i]
            x_refine = self.feat_refine_module[i](x, rois)
            outs = self.refine_head[i](x_refine)
            loss_refine = self.refine_head[i].loss(*outs, gt_bboxes, gt_labels, img_metas, rois=rois,
                                                   gt_bboxes_ignore=gt_bboxes_ignore)
            for name, value in loss_refine.items():
                losses[f'sr{i}.{name}'] = [v * lw for v in value] if 'loss' in name else value
            if i + 1 in range(self.num_refine_stages):
                rois = self.refine_head[i].refine_bboxes(*outs, rois=rois)
        return losses

    @torch.no_grad()
    def simple_test(self, img):
        boxes, scores = self.dense_test(img)
        cfg = self.test_cfg
        return multiclass_nms_rotated_batch(boxes, scores, cfg['score_thr'], cfg['nms'], cfg['max_per_img'],
                                            hint=self.refine_head[-1].nms_hint)

    @torch.no_grad()
    def dense_test(self, img):
        """Network + box decoding: every shape is static and nothing synchronises with the host, so the
        whole of it can be captured in a HIP graph (GraphedDense); the NMS that follows is not (its
        workspace is sized from the candidate counts)."""
        x = self.extract_feat(img)
        cls, reg = self.bbox_head(x)
        rois = self.bbox_head.filter_bboxes(cls, reg)
        for i in range(self.num_refine_stages):
            x_refine = self.feat_refine_module[i](x, rois)
            cls, reg = self.refine_head[i](x_refine)
            if i + 1 < self.num_refine_stages:
                rois = self.refine_head[i].refine_bboxes(cls, reg, rois)
        return self.refine_head[-1].decode_bboxes(cls, reg, img.shape[-2:], self.test_cfg, rois=rois)


class GraphedDense:
    """``model.dense_test(img)`` captured once in a HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm)
    and replayed per step: the ~330 launches of backbone, neck, heads, FRM and the custom ops
    (r3det_filter_bboxes, r3det_feature_refine_*: they enqueue on the current stream, so stream capture
    records them) become one graph launch.  Input and outputs live in static buffers; shapes are fixed
    at capture.  Warm-up runs first so that MIOpen's algorithm search and the library's one-time
    attribute calls happen outside the capture."""

    def __init__(self, model, example, warmup=3):
        self.model = model
        self.static_in = example.clone(memory_format=torch.preserve_format)
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                model.dense_test(self.static_in)
        torch.cuda.current_stream(example.device).wait_stream(side)
        torch.cuda.synchronize(example.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = model.dense_test(self.static_in)

    @torch.no_grad()
    def __call__(self, img):
        if img.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(img)
        self.graph.replay()
        return self.static_out

    @torch.no_grad()
    def simple_test(self, img):
        boxes, scores = self(img)
        cfg = self.model.test_cfg
        return multiclass_nms_rotated_batch(boxes, scores, cfg['score_thr'], cfg['nms'], cfg['max_per_img'],
                                            hint=self.model.refine_head[-1].nms_hint)


def build_detector(cfg):
    """dict(type='R3Det' | 'RRetinaNet', ...) -> module (subset of mmdet's build_detector)."""
    cfg = dict(cfg)
    typ = cfg.pop('type')
    return {'R3Det': R3Det, 'RRetinaNet': RRetinaNet}[typ](**cfg)


@torch.no_grad()
def calibrate_score_bias(model, img, frac=0.01):
    """Synthetic-weight runs give sigmoid ~ 0.01 < score_thr everywhere, i.e.
    an empty NMS pool.  Shift the last head's classification bias so that `frac` of the
    (position, class) scores exceed score_thr -- the sparsity of a trained detector."""
    head = model.refine_head[-1] if isinstance(model, R3Det) else model.bbox_head
    x = model.extract_feat(img)
    if isinstance(model, R3Det):
        cls, reg = model.bbox_head(x)
        rois = model.bbox_head.filter_bboxes(cls, reg)
        x = model.feat_refine_module[-1](x, rois)
    cls, _ = head(x)
    logits = torch.cat([c.flatten() for c in cls])
    k = max(1, int(logits.numel() * frac))
    kth = logits.float().topk(k)[0][-1]
    thr = model.test_cfg['score_thr']
    target = torch.log(torch.tensor(thr / (1 - thr)))
    head.retina_cls.bias.add_(float(target - kth) + 1e-3)
