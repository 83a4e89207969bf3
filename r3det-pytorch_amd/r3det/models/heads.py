"""Rotated RetinaNet heads (models/dense_heads/rotate_retina_head.py,
rotate_retina_refine_head.py, rotate_anchor_head.py) -- inference side + box selection.

The per-image / per-level Python loops of the reference are batched over images here (one
topk / gather / decode per level for the whole batch); the results per image are identical.
"""
import math

import torch
import torch.nn as nn

from ..core.post_processing import CapacityHint, multiclass_nms_rotated_batch
from ..ops import fr_boxes
from .backbone import ConvModule
from .coder import delta2bbox_v1


def level_anchors(featmap_size, stride, device, octave_base_scale=4, scales_per_octave=3,
                  ratios=(1.0, 0.5, 2.0)):
    """RAnchorGenerator.single_level_grid_priors (core/anchor/ranchor_generator.py:11-39) on
    top of mmdet's AnchorGenerator: (H*W*9, 5) as (cx, cy, w, h, 0), position-major."""
    H, W = featmap_size
    scales = torch.tensor([octave_base_scale * 2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
    r = torch.tensor(ratios)
    h_r, w_r = r.sqrt(), 1 / r.sqrt()
    ws = (stride * w_r[:, None] * scales[None, :]).reshape(-1)
    hs = (stride * h_r[:, None] * scales[None, :]).reshape(-1)
    ys, xs = torch.meshgrid(torch.arange(H) * stride, torch.arange(W) * stride, indexing='ij')
    A = ws.numel()
    a = torch.zeros(H * W, A, 5)
    a[:, :, 0] = xs.reshape(-1, 1)
    a[:, :, 1] = ys.reshape(-1, 1)
    a[:, :, 2] = ws
    a[:, :, 3] = hs
    return a.reshape(-1, 5).to(device)


class RRetinaHead(nn.Module):
    """4 x (3x3 conv + ReLU) towers, 9 anchors / position (rotate_retina_head.py:51-115)."""

    def __init__(self, num_classes=15, in_channels=256, stacked_convs=4, feat_channels=256, num_anchors=9,
                 strides=(8, 16, 32, 64, 128), test_cfg=None):
        super().__init__()
        self.num_classes, self.num_anchors, self.strides = num_classes, num_anchors, strides
        self.cls_out_channels = num_classes  # use_sigmoid_cls
        self.test_cfg = test_cfg or dict(nms_pre=2000, min_bbox_size=0, score_thr=0.05,
                                         nms=dict(iou_thr=0.1), max_per_img=2000)
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
        key = (tuple(map(tuple, featmap_sizes)), str(device))
        if key not in self._anchor_cache:
            self._anchor_cache[key] = [level_anchors(fs, s, device) for fs, s in zip(featmap_sizes, self.strides)]
        return self._anchor_cache[key]

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
        HIP graph together with the network."""
        cfg = cfg or self.test_cfg
        N = cls_scores[0].size(0)
        A, C = self.num_anchors, self.cls_out_channels
        if rois is None:
            anchors = self.anchors([c.shape[-2:] for c in cls_scores], cls_scores[0].device)
            lvl_anchors = [a[None].expand(N, -1, -1) for a in anchors]
        else:
            lvl_anchors = [torch.stack([rois[i][l] for i in range(N)]) for l in range(len(cls_scores))]
        nms_pre = cfg.get('nms_pre', -1)
        boxes_l, scores_l = [], []
        for cls, reg, anc in zip(cls_scores, bbox_preds, lvl_anchors):
            scores = cls.permute(0, 2, 3, 1).reshape(N, -1, C).sigmoid()
            reg = reg.permute(0, 2, 3, 1).reshape(N, -1, 5)
            if 0 < nms_pre < scores.shape[1]:
                top = scores.max(dim=2)[0].topk(nms_pre, dim=1)[1]             # (N, nms_pre)
                anc = anc.gather(1, top[..., None].expand(-1, -1, 5))
                reg = reg.gather(1, top[..., None].expand(-1, -1, 5))
                scores = scores.gather(1, top[..., None].expand(-1, -1, C))
            boxes_l.append(delta2bbox_v1(anc, reg, max_shape=img_shape))
            scores_l.append(scores)
        boxes = torch.cat(boxes_l, 1)
        scores = torch.cat(scores_l, 1)
        scores = torch.cat([scores, scores.new_zeros(N, scores.shape[1], 1)], 2)  # dummy background
        return boxes, scores


class RRetinaRefineHead(RRetinaHead):
    """Same towers, one (pseudo) anchor per position: the previous stage's boxes
    (rotate_retina_refine_head.py:20-196)."""

    def __init__(self, num_classes=15, in_channels=256, stacked_convs=4, feat_channels=256,
                 strides=(8, 16, 32, 64, 128), test_cfg=None):
        super().__init__(num_classes, in_channels, stacked_convs, feat_channels, 1, strides, test_cfg)

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
