"""R3Det / RRetinaNet detectors: inference (models/detectors/r3det.py:112-143, rretinanet.py:23-46)
and the training step (r3det.py:75-110; mmdet's SingleStageDetector.forward_train for RRetinaNet)."""
from collections import OrderedDict

import torch
import torch.nn as nn

from ..core.post_processing import multiclass_nms_rotated_batch
from ..ops import FeatureRefineModule
from ..registry import BACKBONES, DETECTORS, HEADS, NECKS, build_from
from .backbone import FPN, ResNet50
from .heads import RRetinaHead, RRetinaRefineHead, _cfg_dict

from .heads import S0_TRAIN_CFG, SR_TRAIN_CFG  # noqa: E402

if 'ResNet' not in BACKBONES:
    BACKBONES.register_module(name='ResNet', module=ResNet50)
if 'FPN' not in NECKS:
    NECKS.register_module(module=FPN)

# configs/r3det/r3det_r50_fpn_1x_dota_v1.py:8-25: what the constructors build when given no backbone / neck dict
BACKBONE_CFG = dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                    zero_init_residual=False, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True,
                    style='pytorch')
NECK_CFG = dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                add_extra_convs='on_input', num_outs=5)

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
    oriented GT (assign_by_circumhbbox=None).  Keywords as mmdet's SingleStageDetector (models/detectors/
    rretinanet.py:9-21): ``RRetinaNet(backbone, neck, bbox_head, train_cfg, test_cfg, pretrained, init_cfg)``; every
    argument defaults to that config."""

    def __init__(self, backbone=None, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 init_cfg=None, num_classes=15):
        super().__init__()
        self.test_cfg = _cfg_dict(test_cfg) if test_cfg is not None else dict(TEST_CFG)
        self.train_cfg = _cfg_dict(train_cfg) if train_cfg is not None else dict(S0_TRAIN_CFG)
        self.backbone = build_from(BACKBONES, _cfg_dict(backbone or BACKBONE_CFG), pretrained=pretrained)
        self.neck = build_from(NECKS, _cfg_dict(neck or NECK_CFG))
        head = _cfg_dict(bbox_head) if bbox_head is not None else dict(
            type='RRetinaHead', num_classes=num_classes, in_channels=256, assign_by_circumhbbox=None,
            loss_bbox=dict(type='L1Loss', loss_weight=1.0))  # (:49 of that config)
        head.update(train_cfg=self.train_cfg, test_cfg=self.test_cfg)
        self.bbox_head = build_from(HEADS, head)

    def extract_feat(self, img):
        return self.neck(self.backbone(img))

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        x = self.extract_feat(img)
        cls, reg = self.bbox_head(x)
        return self.bbox_head.loss(cls, reg, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)

    @torch.no_grad()
    def simple_test(self, img, img_metas=None, rescale=False):
        """``simple_test(img, img_metas, rescale=False)`` (rretinanet.py:23-46): with ``img_metas`` the reference's return
        value (per image a list over classes of (k, 6) ndarrays); without, (dets, labels) tensors per image."""
        boxes, scores = self.dense_test(img)
        return _test_results(self.bbox_head, boxes, scores, self.test_cfg, img_metas, rescale, self.bbox_head.nms_hint)

    @torch.no_grad()
    def dense_test(self, img):
        """Network + decoding + per-level pool (static shapes, no host synchronisation): what GraphedDense /
        GraphedStep record; the multiclass NMS follows (rretinanet.py:23-46)."""
        x = self.extract_feat(img)
        cls, reg = self.bbox_head(x)
        return self.bbox_head.decode_bboxes(cls, reg, img.shape[-2:], self.test_cfg)


def _test_results(head, boxes, scores, cfg, img_metas, rescale, hint):
    """The tail of the reference's ``simple_test`` (models/detectors/r3det.py:136-143, rretinanet.py:38-46) on the dense
    outputs: ``rescale`` divides cx, cy, w, h by the image's ``scale_factor`` BEFORE the NMS (``_get_bboxes_single``,
    rotate_anchor_head.py:657-660: the angle is not rescaled); with ``img_metas`` the detections leave as the
    reference returns them -- per image a list over the classes of (k, 6) float32 ndarrays (``rbbox2result``); without
    (this package's callers: bench, tests) as (dets, labels) tensors on the device."""
    if rescale:
        if img_metas is None:
            raise ValueError('rescale=True needs img_metas with scale_factor')
        sf = torch.stack([boxes.new_tensor(m['scale_factor']).expand(4) if torch.as_tensor(m['scale_factor']).numel() == 1
                          else boxes.new_tensor(m['scale_factor']).reshape(4) for m in img_metas])
        boxes = boxes.clone()
        boxes[..., :4] = boxes[..., :4] / sf[:, None, :]
    res = multiclass_nms_rotated_batch(boxes, scores, cfg['score_thr'], cfg['nms'], cfg['max_per_img'], hint=hint)
    if img_metas is None:
        return res
    from ..core.bbox.rtransforms import rbbox2result
    return [rbbox2result(d, lab, head.num_classes) for d, lab in res]


class R3Det(_Detector):
    """num_refine_stages x (FeatureRefineModule -> RRetinaRefineHead) after the base head;
    state-dict names follow the reference (backbone / neck / bbox_head / feat_refine_module.i /
    refine_head.i)."""

    def __init__(self, num_refine_stages=1, backbone=None, neck=None, bbox_head=None, frm_cfgs=None, refine_heads=None,
                 train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None, num_classes=15):
        """Keywords as the reference's (models/detectors/r3det.py:16-52): the ``model = dict(...)`` of
        configs/r3det/r3det_r50_fpn_1x_dota_v1.py:6-104 builds this class through ``build_detector`` unchanged;
        every argument defaults to that config's value."""
        super().__init__()
        self.test_cfg = _cfg_dict(test_cfg) if test_cfg is not None else dict(TEST_CFG)
        self.train_cfg = _cfg_dict(train_cfg) if train_cfg is not None else dict(R3DET_TRAIN_CFG)
        if len(self.train_cfg['sr']) < num_refine_stages:
            self.train_cfg['sr'] = list(self.train_cfg['sr']) * num_refine_stages
            self.train_cfg['stage_loss_weights'] = list(self.train_cfg['stage_loss_weights']) * num_refine_stages
        self.num_refine_stages = num_refine_stages
        self.backbone = build_from(BACKBONES, _cfg_dict(backbone or BACKBONE_CFG), pretrained=pretrained)
        self.neck = build_from(NECKS, _cfg_dict(neck or NECK_CFG))
        # the base head assigns on the circumscribed horizontal box of the GT (its default 'v1',
        # rotate_anchor_head.py:47,220-224); the refine heads on the oriented GT (config :58)
        head = _cfg_dict(bbox_head) if bbox_head is not None else dict(type='RRetinaHead', num_classes=num_classes,
                                                                       in_channels=256)
        head.update(train_cfg=self.train_cfg['s0'], test_cfg=self.test_cfg)
        self.bbox_head = build_from(HEADS, head)
        frm_cfgs = frm_cfgs or [dict(in_channels=256, featmap_strides=[8, 16, 32, 64, 128])] * num_refine_stages
        if refine_heads is None:
            refine_heads = [dict(type='RRetinaRefineHead', num_classes=num_classes, in_channels=256,
                                 assign_by_circumhbbox=None)] * num_refine_stages
        if len(frm_cfgs) != num_refine_stages or len(refine_heads) != num_refine_stages:
            raise ValueError('frm_cfgs and refine_heads need one entry per refinement stage')
        self.feat_refine_module = nn.ModuleList(FeatureRefineModule(**_cfg_dict(c)) for c in frm_cfgs)
        self.refine_head = nn.ModuleList()
        for i, rh in enumerate(refine_heads):
            rh = _cfg_dict(rh)
            rh.update(train_cfg=self.train_cfg['sr'][i], test_cfg=self.test_cfg)
            self.refine_head.append(build_from(HEADS, rh))
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
    def simple_test(self, img, img_metas=None, rescale=False):
        """``simple_test(img, img_meta, rescale=False)`` (models/detectors/r3det.py:112-143): with ``img_metas`` the
        reference's return value (per image a list over classes of (k, 6) ndarrays); without, (dets, labels) tensors."""
        boxes, scores = self.dense_test(img)
        return _test_results(self.refine_head[-1], boxes, scores, self.test_cfg, img_metas, rescale,
                             self.refine_head[-1].nms_hint)

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
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):  # (RCCL's watchdog thread may query events meanwhile)
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


class GraphedStep:
    """The WHOLE inference step of a detector -- network, box decoding, per-level pool, multiclass NMS and the padded
    result -- captured in ONE HIP graph (round 5; GraphedDense stops in front of the NMS because
    ``multiclass_nms_rotated_batch`` reads the kept counts on the host).  The NMS is ``PaddedNms`` with a fixed
    candidate capacity: nothing in the step looks at a count, the result is the (B, max_per_img + 1, 7) buffer
    ``dist_infer.gather_padded`` sends (rows of [cx, cy, w, h, theta, score, label], the count in the extra row).

    ``step(img)`` replays the graph and returns that buffer.  The capacity's overflow flags are copied to pinned memory
    behind every replay and looked at ``lag`` steps late (default 2: the host never waits for the step it has just
    enqueued, so it stays one whole step ahead of the GPU): the rare step whose pool outgrew the capacity is reported
    by a LATER call -- ``redo`` = how many of the most recent steps to run again (0 almost always), after the graph has
    been recorded again with twice the capacity.  A loop over ``step`` ends with ``flush()`` (the last ``lag`` steps
    have not been looked at yet).  ``simple_test`` checks its own step before it returns.  Reference semantics:
    models/detectors/r3det.py:112-143."""

    def __init__(self, model, example, cap=None, warmup=3, lag=2):
        from ..core.post_processing import PaddedNms
        self.model = model
        self.lag = max(1, int(lag))
        self.static_in = example.clone(memory_format=torch.preserve_format)
        cfg = model.test_cfg
        dev = example.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                boxes, scores = model.dense_test(self.static_in)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        B, n = boxes.shape[:2]
        K = scores.size(2) - 1
        if cap is None:  # the warm-up pool's largest image, with room (a host read, once, outside the steps)
            m = int((scores[..., :-1] > cfg['score_thr']).flatten(1).sum(1).max().item())
            cap = max(1024, int(m * 1.3))
        self.nms = PaddedNms(B, n, K, cfg['score_thr'], cfg['nms'], cfg['max_per_img'], cap, dev)
        self._record()

    def _record(self):
        dev = self.static_in.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # (the library's one-time attribute calls happen outside the capture)
            self.nms(*self.model.dense_test(self.static_in))
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):  # (RCCL's watchdog thread may query events meanwhile)
            boxes, scores = self.model.dense_test(self.static_in)
            self.static_out = self.nms(boxes.contiguous(), scores.contiguous())

    def _regrow(self):
        """An overflow was seen: every step still pending ran on the same short capacity -- drop their flags (the
        caller repeats them all), double the capacity, record the graph again.  -> how many steps to repeat."""
        again = 1 + self.nms.pending()
        self.nms.check(1)
        self.nms.grow()
        self._record()
        return again

    @torch.no_grad()
    def step(self, img):
        """-> (out (B, max_per_img + 1, 7), redo).  ``redo`` > 0: one of the last ``redo`` steps BEFORE this one
        overflowed the candidate capacity (its result covered its first ``cap`` candidates only); the graph now has
        twice the capacity -- run those ``redo`` steps again."""
        redo = self._regrow() if self.nms.check(self.lag) else 0
        if img.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(img)
        self.graph.replay()
        self.nms.post_flags()
        return self.static_out, redo

    @torch.no_grad()
    def flush(self):
        """After the last ``step`` of a loop: wait for the steps not looked at yet.  -> ``redo`` as ``step`` returns it
        (how many of the most recent steps overflowed the capacity's graph and have to be run again)."""
        n = self.nms.pending()
        if self.nms.check(1):
            self.nms.grow()
            self._record()
            return n
        return 0

    @torch.no_grad()
    def simple_test(self, img):
        """The reference's per-image (dets, labels) lists (reads the counts: one host synchronisation).  The counts
        come back together with this step's overflow flags: a pool beyond the capacity is run again on a graph with
        room for it BEFORE anything is returned, so the lists always equal ``multiclass_nms_rotated``'s."""
        self.flush()                      # (earlier ``step`` calls: their redo is the ``step`` caller's business)
        if img.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(img)
        while True:
            self.graph.replay()
            _, over = self.nms.read()
            if not any(over):
                return self.nms.lists()
            self.nms.grow()
            self._record()


for _c in (R3Det, RRetinaNet):
    if _c.__name__ not in DETECTORS:
        DETECTORS.register_module(module=_c)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet's build_detector on this package's registries: ``cfg`` is the ``model = dict(type='R3Det' |
    'RRetinaNet', backbone=dict(...), neck=dict(...), bbox_head=dict(...), ...)`` of a reference config (a dict, an
    mmcv Config node, anything with keys())."""
    cfg = _cfg_dict(cfg)
    if train_cfg is not None:
        cfg['train_cfg'] = train_cfg
    if test_cfg is not None:
        cfg['test_cfg'] = test_cfg
    return build_from(DETECTORS, cfg)


@torch.no_grad()
def calibrate_score_bias(model, img, frac=0.01, per_class=False):
    """Synthetic-weight runs give sigmoid ~ 0.01 < score_thr everywhere, i.e.
    an empty NMS pool.  Shift the last head's classification bias so that `frac` of the
    (position, class) scores exceed score_thr -- the sparsity of a trained detector.

    per_class: one shift PER CLASS, every class the same share of the candidates ("15 labels spread", the
    DOTA-shaped pool; configs/r3det/r3det_r50_fpn_1x_dota_v1.py:99-104).  One common shift leaves ~90 % of a
    random-weight model's candidates in a single class -- an artefact of the random head, not of the dataset."""
    head = model.refine_head[-1] if isinstance(model, R3Det) else model.bbox_head
    x = model.extract_feat(img)
    if isinstance(model, R3Det):
        cls, reg = model.bbox_head(x)
        rois = model.bbox_head.filter_bboxes(cls, reg)
        x = model.feat_refine_module[-1](x, rois)
    cls, _ = head(x)
    thr = model.test_cfg['score_thr']
    target = torch.log(torch.tensor(thr / (1 - thr)))
    if per_class:
        C = head.cls_out_channels
        A = cls[0].size(1) // C
        per = [c.reshape(c.size(0), A, C, -1).permute(2, 0, 1, 3).reshape(C, -1) for c in cls]  # channel = a * C + c
        logits = torch.cat(per, 1).float()                                                       # (C, everything else)
        k = max(1, int(logits.size(1) * frac))
        kth = logits.topk(k, dim=1)[0][:, -1]
        shift = (target.to(kth.device) - kth + 1e-3).to(head.retina_cls.bias.dtype)
        head.retina_cls.bias.add_(shift.repeat(A))
        return
    logits = torch.cat([c.flatten() for c in cls])
    k = max(1, int(logits.numel() * frac))
    kth = logits.float().topk(k)[0][-1]
    head.retina_cls.bias.add_(float(target - kth) + 1e-3)
