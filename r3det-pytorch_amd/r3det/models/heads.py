"""Rotated RetinaNet heads (models/dense_heads/rotate_retina_head.py,
rotate_retina_refine_head.py, rotate_anchor_head.py): inference side, box selection for the FR
sampler, and the training side (targets from the fused assignment, focal + smooth-L1 losses).

The per-image / per-level Python loops of the reference are batched over images here (one
topk / gather / decode per level for the whole batch); the results per image are identical.
The training targets are built with masks instead of index lists (no ``nonzero``: nothing in
``loss`` synchronises with the host), which gives the same tensors as the reference's
``_get_targets_single`` (rotate_anchor_head.py:172-277) for its PseudoSampler configuration.
"""
import math

import torch
import torch.nn as nn

from ..core.anchor import RAnchorGenerator, ranchor_inside_flags
from ..core.bbox.assigners import MaxIoUAssigner
from ..core.bbox.coder import DeltaXYWHAOBBoxCoder
from ..core.bbox.rtransforms import obb2hbb
from ..core.post_processing import CapacityHint, multiclass_nms_rotated_batch
from ..ops import fr_boxes
from .backbone import ConvModule
from .coder import delta2bbox_v1
from .losses import build_loss
from ..registry import BBOX_ASSIGNERS, BBOX_CODERS, HEADS, PRIOR_GENERATORS, build_from

for _n, _c in (('MaxIoUAssigner', MaxIoUAssigner),):
    if _n not in BBOX_ASSIGNERS:
        BBOX_ASSIGNERS.register_module(name=_n, module=_c)
if 'DeltaXYWHAOBBoxCoder' not in BBOX_CODERS:
    BBOX_CODERS.register_module(module=DeltaXYWHAOBBoxCoder)
from ..core.anchor import PseudoAnchorGenerator  # noqa: E402
for _c in (RAnchorGenerator, PseudoAnchorGenerator):
    if _c.__name__ not in PRIOR_GENERATORS:
        PRIOR_GENERATORS.register_module(module=_c)

S0_TRAIN_CFG = dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0,
                                  ignore_iof_thr=-1, iou_calculator=dict(type='RBboxOverlaps2D_v1')),
                    allowed_border=-1, pos_weight=-1, debug=False)
SR_TRAIN_CFG = dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.6, neg_iou_thr=0.5, min_pos_iou=0,
                                  ignore_iof_thr=-1, iou_calculator=dict(type='RBboxOverlaps2D_v1')),
                    allowed_border=-1, pos_weight=-1, debug=False)


def build_assigner(cfg):
    return build_from(BBOX_ASSIGNERS, cfg)  # (only MaxIoUAssigner is registered: the rotated configs use no other)


def _cfg_dict(cfg):
    """A (possibly attribute-style) config mapping -> plain nested dicts."""
    if hasattr(cfg, 'keys'):
        return {k: _cfg_dict(cfg[k]) for k in cfg.keys()}
    if isinstance(cfg, (list, tuple)):
        return type(cfg)(_cfg_dict(v) for v in cfg)
    return cfg


def level_anchors(featmap_size, stride, device, octave_base_scale=4, scales_per_octave=3,
                  ratios=(1.0, 0.5, 2.0)):
    """One level of RAnchorGenerator (core/anchor/ranchor_generator.py): (H*W*9, 5) as
    (cx, cy, w, h, 0), position-major."""
    gen = RAnchorGenerator([stride], list(ratios), octave_base_scale=octave_base_scale,
                           scales_per_octave=scales_per_octave)
    return gen.single_level_grid_priors(featmap_size, 0, device=device)


class RRetinaHead(nn.Module):
    """4 x (3x3 conv + ReLU) towers, 9 anchors / position (rotate_retina_head.py:51-115).

    Constructor keywords are the reference's (rotate_retina_head.py:29-49 + RAnchorHead, rotate_anchor_head.py:33-98):
    ``RRetinaHead(num_classes, in_channels, stacked_convs=4, feat_channels=256, anchor_generator=dict(type=
    'RAnchorGenerator', ...), bbox_coder=dict(...), loss_cls=dict(...), loss_bbox=dict(...), train_cfg=..., test_cfg=
    ..., assign_by_circumhbbox='v1')`` -- so the ``bbox_head`` dict of configs/r3det/r3det_r50_fpn_1x_dota_v1.py
    builds it unchanged (registry name ``RRetinaHead``).  Every argument defaults to that config's value;
    ``strides`` / ``num_anchors`` are shorthands for the default anchor generator on other strides."""

    def __init__(self, num_classes=15, in_channels=256, stacked_convs=4, feat_channels=256, num_anchors=None,
                 strides=None, test_cfg=None, train_cfg=None, assign_by_circumhbbox='v1', bbox_coder=None,
                 loss_cls=None, loss_bbox=None, conv_cfg=None, norm_cfg=None, anchor_generator=None,
                 reg_decoded_bbox=False):
        super().__init__()
        if conv_cfg is not None or norm_cfg is not None or reg_decoded_bbox:
            raise NotImplementedError('RRetinaHead: conv_cfg / norm_cfg / reg_decoded_bbox are not used by the '
                                      'rotated configs and not restated')
        if anchor_generator is None:
            anchor_generator = self._default_anchor_generator(strides or (8, 16, 32, 64, 128))
        self.anchor_generator = build_from(PRIOR_GENERATORS, anchor_generator)
        self.strides = tuple(s[0] for s in self.anchor_generator.strides)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        if num_anchors is not None and num_anchors != self.num_anchors:
            raise ValueError(f'num_anchors={num_anchors} contradicts the anchor generator ({self.num_anchors})')
        self.num_classes, self.in_channels, self.feat_channels = num_classes, in_channels, feat_channels
        self.stacked_convs = stacked_convs
        loss_cls = _cfg_dict(loss_cls or dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25,
                                              loss_weight=1.0))
        if not loss_cls.get('use_sigmoid', False):
            raise NotImplementedError('softmax classification heads are not used by the rotated configs')
        self.use_sigmoid_cls = True
        self.cls_out_channels = num_classes  # use_sigmoid_cls
        self.test_cfg = _cfg_dict(test_cfg) if test_cfg is not None else dict(
            nms_pre=2000, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
        # training side (rotate_anchor_head.py:33-98); defaults = configs/r3det/r3det_r50_fpn_1x_dota_v1.py
        self.train_cfg = _cfg_dict(train_cfg) if train_cfg is not None else dict(self._default_train_cfg())
        self.assign_by_circumhbbox = assign_by_circumhbbox
        self.assigner = build_assigner(self.train_cfg['assigner'])
        self.bbox_coder = build_from(BBOX_CODERS, _cfg_dict(bbox_coder or dict(type='DeltaXYWHAOBBoxCoder')))
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(_cfg_dict(loss_bbox or dict(type='SmoothL1Loss', beta=0.11, loss_weight=1.0)))
        num_anchors = self.num_anchors
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        for i in range(stacked_convs):
            c = in_channels if i == 0 else feat_channels
            self.cls_convs.append(ConvModule(c, feat_channels, 3, padding=1, act=True))
            self.reg_convs.append(ConvModule(c, feat_channels, 3, padding=1, act=True))
        self.retina_cls = nn.Conv2d(feat_channels, num_anchors * num_classes, 3, padding=1)
        self.retina_reg = nn.Conv2d(feat_channels, num_anchors * 5, 3, padding=1)
        self._anchor_cache = {}
        self.nms_hint = CapacityHint()  # this head's own workspace-size memory for the batched NMS
        self.init_weights()

    @staticmethod
    def _default_anchor_generator(strides):
        return dict(type='RAnchorGenerator', octave_base_scale=4, scales_per_octave=3, ratios=[1.0, 0.5, 2.0],
                    strides=list(strides))

    @staticmethod
    def _default_train_cfg():
        return S0_TRAIN_CFG

    def init_weights(self):
        for m in list(self.cls_convs) + list(self.reg_convs):
            nn.init.normal_(m.conv.weight, 0, 0.01)
            nn.init.constant_(m.conv.bias, 0)
        nn.init.normal_(self.retina_cls.weight, 0, 0.01)
        nn.init.constant_(self.retina_cls.bias, float(-math.log((1 - 0.01) / 0.01)))  # bias_init_with_prob
        nn.init.normal_(self.retina_reg.weight, 0, 0.01)
        nn.init.constant_(self.retina_reg.bias, 0)

    def forward_single(self, x):
        c, r = x, x
        for m in self.cls_convs:
            c = m(c)
        for m in self.reg_convs:
            r = m(r)
        return self.retina_cls(c), self.retina_reg(r)

    def forward(self, feats):
        outs = [self.forward_single(f) for f in feats]
        return [o[0] for o in outs], [o[1] for o in outs]

    def anchors(self, featmap_sizes, device):
        return self.anchor_generator.grid_priors([tuple(fs) for fs in featmap_sizes], device=device)

    def _grid_key(self, anchor_list):
        """Names the anchor list for the assigner's prepared columns when it IS the generator's grid (the anchor head;
        the refine head's anchors are the previous stage's boxes: None)."""
        # (the level tensors are the generator's cached ones, one set per tuple of feature-map SIZES: their addresses tell
        # a 64 x 100 grid from a 100 x 64 one, which the per-level counts do not)
        return ('anchor_grid', id(self.anchor_generator), tuple((int(a.size(0)), int(a.data_ptr())) for a in anchor_list[0]),
                str(anchor_list[0][0].device))

    # ------------------------------------------------------------------ training
    def get_anchors(self, featmap_sizes, img_metas, device):
        """list[img] of list[lvl] anchors (shared) and, per image, whether every anchor is valid /
        the valid flags (rotate_anchor_head.py:143-170).  The flags follow from the padded shape, so
        the all-valid case is decided on the host."""
        lvl = self.anchors(featmap_sizes, device)
        flags = []
        for meta in img_metas:
            h, w = meta['pad_shape'][:2]
            full = all(min(math.ceil(h / s), fh) == fh and min(math.ceil(w / s), fw) == fw
                       for (fh, fw), s in zip(featmap_sizes, self.strides))
            flags.append(None if full else torch.cat(self.anchor_generator.valid_flags(
                [tuple(fs) for fs in featmap_sizes], meta['pad_shape'], device=device)))
        return [lvl for _ in img_metas], flags

    def _targets_single(self, flat_anchors, valid_flags, gt_bboxes, gt_labels, img_meta, gt_bboxes_ignore=None,
                        grid_key=None):
        """rotate_anchor_head.py:172-277 for one image, PseudoSampler configuration: every assigned
        anchor is a positive, every anchor with gt_inds == 0 a negative.  Returns labels (n,),
        label_weights (n,), bbox_targets (n, 5), bbox_weights (n, 5), number of positives (0-dim)."""
        n_all = flat_anchors.size(0)
        inside = None
        border = self.train_cfg.get('allowed_border', -1)
        if valid_flags is not None or border >= 0:
            vf = valid_flags if valid_flags is not None else flat_anchors.new_ones(n_all, dtype=torch.bool)
            inside = ranchor_inside_flags(flat_anchors, vf, img_meta['img_shape'][:2], border)
            if not bool(inside.any()):
                return None  # rotate_anchor_head.py:215-216 (only reached with partial valid flags / a border)
        anchors = flat_anchors if inside is None else flat_anchors[inside]
        gt_bboxes = gt_bboxes.to(flat_anchors.dtype)
        gt_assign = gt_bboxes
        if self.assign_by_circumhbbox is not None and gt_bboxes.size(0) > 0:
            gt_assign = obb2hbb(gt_bboxes, self.assign_by_circumhbbox)
        # (labels are derived below with masks: passing gt_labels would make the assigner run nonzero(), a host sync)
        # (grid_key: the anchors are the generator's grid of these feature-map sizes, the same list in every step and for
        # every image: the assigner prepares its columns once)
        res = self.assigner.assign(anchors, gt_assign, gt_bboxes_ignore, None,
                                   shared_key=grid_key if inside is None else None)
        gt_inds = res.gt_inds
        pos = gt_inds > 0
        n = anchors.size(0)
        if gt_bboxes.size(0) > 0:
            idx = (gt_inds - 1).clamp(min=0)
            enc = self.bbox_coder.encode(anchors, gt_bboxes[idx])
            bbox_targets = torch.where(pos[:, None], enc, torch.zeros_like(enc))
            labels = torch.where(pos, gt_labels[idx], torch.full_like(gt_inds, self.num_classes))
        else:
            bbox_targets = torch.zeros_like(anchors)
            labels = gt_inds.new_full((n,), self.num_classes)
        bbox_weights = pos[:, None].to(anchors.dtype).expand(n, 5)
        label_weights = (gt_inds >= 0).to(anchors.dtype)
        pw = self.train_cfg.get('pos_weight', -1)
        if pw > 0:
            label_weights = torch.where(pos, label_weights.new_tensor(float(pw)), label_weights)
        if inside is not None:  # unmap (:262-272)
            def unmap(t, fill=0):
                out = t.new_full((n_all,) + t.shape[1:], fill)
                out[inside] = t
                return out
            labels, label_weights = unmap(labels, self.num_classes), unmap(label_weights)
            bbox_targets, bbox_weights = unmap(bbox_targets), unmap(bbox_weights)
        return labels, label_weights, bbox_targets, bbox_weights, pos.sum()

    def get_targets(self, anchor_list, valid_flag_list, gt_bboxes_list, img_metas, gt_labels_list,
                    gt_bboxes_ignore_list=None):
        """rotate_anchor_head.py:279-377: per level (N, n_l[, 5]) targets and
        ``num_total_pos = sum_i max(#pos_i, 1)`` (a device scalar); None when an image has no anchor inside
        (:352-353)."""
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
        if gt_bboxes_ignore_list is None:
            gt_bboxes_ignore_list = [None] * len(img_metas)
        grid_key = self._grid_key(anchor_list)
        per_img = [self._targets_single(torch.cat(anchor_list[i]), valid_flag_list[i], gt_bboxes_list[i],
                                        gt_labels_list[i], img_metas[i], gt_bboxes_ignore_list[i], grid_key)
                   for i in range(len(img_metas))]
        if any(r is None for r in per_img):
            return None
        labels, label_w, bbox_t, bbox_w = (torch.stack([r[k] for r in per_img]) for k in range(4))
        num_total_pos = torch.stack([r[4] for r in per_img]).clamp(min=1).sum()
        split = lambda t: list(t.split(num_level_anchors, dim=1))  # noqa: E731  (images_to_levels)
        return split(labels), split(label_w), split(bbox_t), split(bbox_w), num_total_pos

    def loss_single(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, num_total_samples):
        """rotate_anchor_head.py:379-427 (reg_decoded_bbox=False)."""
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
        loss_cls = self.loss_cls(cls_score, labels.reshape(-1), label_weights.reshape(-1),
                                 avg_factor=num_total_samples)
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, 5)
        loss_bbox = self.loss_bbox(bbox_pred, bbox_targets.reshape(-1, 5), bbox_weights.reshape(-1, 5),
                                   avg_factor=num_total_samples)
        return loss_cls, loss_bbox

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        """rotate_anchor_head.py:429-497 -> dict(loss_cls=[per level], loss_bbox=[per level]); the head
        outputs are taken in fp32 (``force_fp32``)."""
        featmap_sizes = [f.shape[-2:] for f in cls_scores]
        assert len(featmap_sizes) == len(self.strides)
        anchor_list, valid_flag_list = self.get_anchors(featmap_sizes, img_metas, cls_scores[0].device)
        targets = self.get_targets(anchor_list, valid_flag_list, gt_bboxes, img_metas, gt_labels, gt_bboxes_ignore)
        if targets is None:
            return None  # rotate_anchor_head.py:469-470
        labels, label_w, bbox_t, bbox_w, num_total_pos = targets
        avg = num_total_pos.to(torch.float32)
        out = [self.loss_single(c.float(), r.float(), la, lw, bt, bw, avg)
               for c, r, la, lw, bt, bw in zip(cls_scores, bbox_preds, labels, label_w, bbox_t, bbox_w)]
        return dict(loss_cls=[o[0] for o in out], loss_bbox=[o[1] for o in out])

    @torch.no_grad()
    def filter_bboxes(self, cls_scores, bbox_preds):
        """Best-scoring anchor per position, decoded (rotate_retina_head.py:117-179).
        Returns list[img][lvl] of (H*W, 5).  One fused launch per level on the device
        (ops/fr_boxes.py); ``filter_bboxes_torch`` is the op-by-op form the tests compare with."""
        if not cls_scores[0].is_cuda:
            return self.filter_bboxes_torch(cls_scores, bbox_preds)
        N = cls_scores[0].size(0)
        anchors = self.anchors([c.shape[-2:] for c in cls_scores], cls_scores[0].device)
        out = [[] for _ in range(N)]
        for cls, reg, anc in zip(cls_scores, bbox_preds, anchors):
            boxes = fr_boxes.filter_bboxes(cls, reg, anc, self.num_anchors, self.cls_out_channels)
            for i in range(N):
                out[i].append(boxes[i])
        return out

    @torch.no_grad()
    def filter_bboxes_torch(self, cls_scores, bbox_preds):
        N = cls_scores[0].size(0)
        anchors = self.anchors([c.shape[-2:] for c in cls_scores], cls_scores[0].device)
        out = [[] for _ in range(N)]
        for cls, reg, anc in zip(cls_scores, bbox_preds, anchors):
            A, C = self.num_anchors, self.cls_out_channels
            cls = cls.permute(0, 2, 3, 1).reshape(N, -1, A, C)
            best = cls.max(dim=-1)[0].argmax(dim=-1)                            # (N, HW)
            reg = reg.permute(0, 2, 3, 1).reshape(N, -1, A, 5)
            idx = best[..., None, None].expand(-1, -1, 1, 5)
            best_pred = reg.gather(2, idx).squeeze(2)                           # (N, HW, 5)
            best_anchor = anc.reshape(1, -1, A, 5).expand(N, -1, -1, -1).gather(2, idx).squeeze(2)
            boxes = delta2bbox_v1(best_anchor, best_pred)
            for i in range(N):
                out[i].append(boxes[i])
        return out

    @torch.no_grad()
    def get_bboxes(self, cls_scores, bbox_preds, img_shape, cfg=None, rois=None):
        """rotate_anchor_head.py:499-675 (+ the refine head's rois-as-anchors variant,
        rotate_retina_refine_head.py:147-196).  Returns [(dets (k,6), labels (k,))] per image."""
        cfg = cfg or self.test_cfg
        boxes, scores = self.decode_bboxes(cls_scores, bbox_preds, img_shape, cfg, rois)
        return multiclass_nms_rotated_batch(boxes, scores, cfg['score_thr'], cfg['nms'], cfg['max_per_img'],
                                            hint=self.nms_hint)

    def decode_bboxes(self, cls_scores, bbox_preds, img_shape, cfg=None, rois=None):
        """The shape-static part of get_bboxes (everything before the NMS, whose sizes depend on the
        scores): (N, n, 5) boxes and (N, n, C + 1) scores.  No host synchronisation: capturable in a
        HIP graph together with the network.  On the device one library call for all levels (r3det_levels_pool:
        sigmoid, per-image top-nms_pre in score order, decoding, straight into the pool arrays; level by level
        through r3det_level_pool / torch when a level is beyond the library's top-k); the op-by-op torch form is
        ``decode_bboxes_torch``."""
        cfg = cfg or self.test_cfg
        nms_pre = cfg.get('nms_pre', -1)
        nms_pre = -1 if nms_pre is None else int(nms_pre)
        if not cls_scores[0].is_cuda or cls_scores[0].dtype != torch.float32:
            return self.decode_bboxes_torch(cls_scores, bbox_preds, img_shape, cfg, rois)
        N = cls_scores[0].size(0)
        A, C = self.num_anchors, self.cls_out_channels
        dev = cls_scores[0].device
        if rois is None:
            lvl_anchors = self.anchors([c.shape[-2:] for c in cls_scores], dev)
        else:
            lvl_anchors = [torch.stack([rois[i][l] for i in range(N)]) for l in range(len(cls_scores))]
        rows = [c.shape[-2] * c.shape[-1] * A for c in cls_scores]
        rows = [min(nms_pre, r) if nms_pre > 0 else r for r in rows]
        n = sum(rows)
        boxes = torch.empty((N, n, 5), dtype=torch.float32, device=dev)
        scores = torch.empty((N, n, C + 1), dtype=torch.float32, device=dev)
        too_big = [0 < nms_pre < c.shape[-2] * c.shape[-1] * A and
                   (nms_pre > fr_boxes.POOL_MAX_K or c.shape[-2] * c.shape[-1] * A > fr_boxes.POOL_MAX_ROWS)
                   for c in cls_scores]
        if not any(too_big) and len(cls_scores) <= fr_boxes.POOL_MAX_LEVELS:
            # all levels in one library call (three launches for the whole pool)
            fr_boxes.levels_pool(list(cls_scores), list(bbox_preds), lvl_anchors, A, C, nms_pre, img_shape, boxes, scores)
            return boxes, scores
        off = 0
        for cls, reg, anc, r in zip(cls_scores, bbox_preds, lvl_anchors, rows):
            L = cls.shape[-2] * cls.shape[-1] * A
            # the library's top-k holds at most POOL_MAX_K winners and POOL_MAX_ROWS keys per image and level
            # (r3_pool.hip); a level beyond that takes the op-by-op form, level by level (a 4096^2 test image at
            # stride 8 has 2.36 M rows)
            if 0 < nms_pre < L and (nms_pre > fr_boxes.POOL_MAX_K or L > fr_boxes.POOL_MAX_ROWS):
                b, sc = self._level_torch(cls, reg, anc if anc.dim() == 3 else anc[None].expand(N, -1, -1), nms_pre,
                                          img_shape)
                boxes[:, off:off + r] = b
                scores[:, off:off + r, :C] = sc
                scores[:, off:off + r, C] = 0
            else:
                fr_boxes.level_pool(cls, reg, anc, A, C, nms_pre, img_shape, boxes, scores, off)
            off += r
        return boxes, scores

    def _level_torch(self, cls, reg, anc, nms_pre, img_shape):
        """One level of rotate_anchor_head.py:626-660 for the whole batch, op by op: (N, r, 5), (N, r, C)."""
        N, C = cls.size(0), self.cls_out_channels
        scores = cls.permute(0, 2, 3, 1).reshape(N, -1, C).sigmoid()
        reg = reg.permute(0, 2, 3, 1).reshape(N, -1, 5)
        if 0 < nms_pre < scores.shape[1]:
            top = scores.max(dim=2)[0].topk(nms_pre, dim=1)[1]             # (N, nms_pre)
            anc = anc.gather(1, top[..., None].expand(-1, -1, 5))
            reg = reg.gather(1, top[..., None].expand(-1, -1, 5))
            scores = scores.gather(1, top[..., None].expand(-1, -1, C))
        return delta2bbox_v1(anc, reg, max_shape=img_shape), scores

    def decode_bboxes_torch(self, cls_scores, bbox_preds, img_shape, cfg=None, rois=None):
        cfg = cfg or self.test_cfg
        N = cls_scores[0].size(0)
        if rois is None:
            anchors = self.anchors([c.shape[-2:] for c in cls_scores], cls_scores[0].device)
            lvl_anchors = [a[None].expand(N, -1, -1) for a in anchors]
        else:
            lvl_anchors = [torch.stack([rois[i][l] for i in range(N)]) for l in range(len(cls_scores))]
        nms_pre = cfg.get('nms_pre', -1)
        nms_pre = -1 if nms_pre is None else int(nms_pre)
        per = [self._level_torch(cls, reg, anc, nms_pre, img_shape)
               for cls, reg, anc in zip(cls_scores, bbox_preds, lvl_anchors)]
        boxes = torch.cat([p[0] for p in per], 1)
        scores = torch.cat([p[1] for p in per], 1)
        scores = torch.cat([scores, scores.new_zeros(N, scores.shape[1], 1)], 2)  # dummy background
        return boxes, scores


class RRetinaRefineHead(RRetinaHead):
    """Same towers, one (pseudo) anchor per position: the previous stage's boxes
    (rotate_retina_refine_head.py:20-196).  Registry name ``RRetinaRefineHead``; keywords as the reference's."""

    @staticmethod
    def _default_train_cfg():
        return SR_TRAIN_CFG

    def __init__(self, num_classes=15, in_channels=256, stacked_convs=4, feat_channels=256, strides=None, test_cfg=None,
                 train_cfg=None, assign_by_circumhbbox=None, anchor_generator=None, **kwargs):
        if anchor_generator is None:  # (rotate_retina_refine_head.py:36-38)
            anchor_generator = dict(type='PseudoAnchorGenerator', strides=list(strides or (8, 16, 32, 64, 128)))
        super().__init__(num_classes, in_channels, stacked_convs, feat_channels, test_cfg=test_cfg, train_cfg=train_cfg,
                         assign_by_circumhbbox=assign_by_circumhbbox, anchor_generator=anchor_generator, **kwargs)
        self.bboxes_as_anchors = None

    def _grid_key(self, anchor_list):
        return None  # (the anchors are the previous stage's boxes: different in every step)

    def get_anchors(self, featmap_sizes, img_metas, device):
        """The previous stage's boxes are the anchors (rotate_retina_refine_head.py:99-125); the
        PseudoAnchorGenerator only supplies valid flags (one per position)."""
        anchor_list = [[b.detach() for b in per_img] for per_img in self.bboxes_as_anchors]
        flags = []
        for meta in img_metas:
            h, w = meta['pad_shape'][:2]
            full = all(min(math.ceil(h / s), fh) == fh and min(math.ceil(w / s), fw) == fw
                       for (fh, fw), s in zip(featmap_sizes, self.strides))
            if full:
                flags.append(None)
            else:
                per = self.anchor_generator.valid_flags([tuple(fs) for fs in featmap_sizes], meta['pad_shape'],
                                                        device=device)
                flags.append(torch.cat(per))  # (PseudoAnchorGenerator: one flag per position)
        return anchor_list, flags

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, rois=None, gt_bboxes_ignore=None):
        """rotate_retina_refine_head.py:127-145."""
        assert rois is not None
        self.bboxes_as_anchors = rois
        return super().loss(cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore)

    @torch.no_grad()
    def refine_bboxes(self, cls_scores, bbox_preds, rois):
        if not bbox_preds[0].is_cuda:
            return self.refine_bboxes_torch(cls_scores, bbox_preds, rois)
        N = cls_scores[0].size(0)
        out = [[] for _ in range(N)]
        for l, reg in enumerate(bbox_preds):
            ref = fr_boxes.refine_bboxes(reg, torch.stack([rois[i][l] for i in range(N)]))
            for i in range(N):
                out[i].append(ref[i])
        return out

    @torch.no_grad()
    def refine_bboxes_torch(self, cls_scores, bbox_preds, rois):
        N = cls_scores[0].size(0)
        out = [[] for _ in range(N)]
        for l, reg in enumerate(bbox_preds):
            reg = reg.permute(0, 2, 3, 1).reshape(N, -1, 5)
            anc = torch.stack([rois[i][l] for i in range(N)])
            ref = delta2bbox_v1(anc, reg)
            for i in range(N):
                out[i].append(ref[i])
        return out


for _c in (RRetinaHead, RRetinaRefineHead):
    if _c.__name__ not in HEADS:
        HEADS.register_module(module=_c)
