"""DeltaXYWHAOBBoxCoder, angle version v1 (core/bbox/coder/delta_xywha_rbbox_coder.py:11-211):
elementwise torch, means 0 / stds 1 in the shipped configs.  On the device the decoding used by the
FR box producers is fused into ``r3det_filter_bboxes`` (csrc/r3_boxes.hip); these functions are the
host-side form (loss targets, final decoding) and what tests/golden/heads.npz pins that kernel to.
"""
import math

import torch


_consts = {}


def _const(like, values):
    """means / stds as a device tensor, made once per (device, dtype, values): a host-to-device copy per
    call would also keep the decoding out of a HIP graph capture."""
    key = (like.device, like.dtype, tuple(float(v) for v in values))
    t = _consts.get(key)
    if t is None:
        t = _consts[key] = like.new_tensor(values)
    return t


def delta2bbox_v1(rois, deltas, means=(0., 0., 0., 0., 0.), stds=(1., 1., 1., 1., 1.), max_shape=None,
                  wh_ratio_clip=16 / 1000, add_ctr_clamp=False, ctr_clamp=32):
    """rois (..., 5), deltas (..., 5) or (..., k*5) -> boxes shaped like ``deltas``; dw/dh clamped to
    |log(wh_ratio_clip)|, centres clamped to the image when max_shape=(H, W) is given (:142-211)."""
    shape = deltas.shape
    if deltas.size(-1) != 5:
        deltas = deltas.reshape(*shape[:-1], -1, 5)
        rois = rois.unsqueeze(-2)
    m, s = _const(deltas, means), _const(deltas, stds)
    d = deltas * s + m
    max_ratio = abs(math.log(wh_ratio_clip))
    dxw = rois[..., 2] * d[..., 0]
    dyh = rois[..., 3] * d[..., 1]
    if add_ctr_clamp:  # YOLOF only (:188-192)
        dxw = dxw.clamp(min=-ctr_clamp, max=ctr_clamp)
        dyh = dyh.clamp(min=-ctr_clamp, max=ctr_clamp)
        dw = d[..., 2].clamp(max=max_ratio)
        dh = d[..., 3].clamp(max=max_ratio)
    else:
        dw = d[..., 2].clamp(min=-max_ratio, max=max_ratio)
        dh = d[..., 3].clamp(min=-max_ratio, max=max_ratio)
    gw = rois[..., 2] * dw.exp()
    gh = rois[..., 3] * dh.exp()
    gx = rois[..., 0] + dxw
    gy = rois[..., 1] + dyh
    ga = rois[..., 4] + d[..., 4]
    if max_shape is not None:
        gx = gx.clamp(min=0, max=max_shape[1] - 1)
        gy = gy.clamp(min=0, max=max_shape[0] - 1)
    return torch.stack([gx, gy, gw, gh, ga], dim=-1).reshape(shape)


def bbox2delta_v1(proposals, gt, means=(0., 0., 0., 0., 0.), stds=(1., 1., 1., 1., 1.)):
    """Inverse of delta2bbox_v1 (:104-139)."""
    p, g = proposals.float(), gt.float()
    d = torch.stack([(g[..., 0] - p[..., 0]) / p[..., 2], (g[..., 1] - p[..., 1]) / p[..., 3],
                     torch.log(g[..., 2] / p[..., 2]), torch.log(g[..., 3] / p[..., 3]),
                     g[..., 4] - p[..., 4]], dim=-1)
    return (d - _const(d, means)) / _const(d, stds)


class DeltaXYWHAOBBoxCoder:
    """``dict(type='DeltaXYWHAOBBoxCoder', target_means=..., target_stds=...)`` (:11-100).  Only the
    angle version of the BASELINE configs ('v1') is restated; 'v2' / 'v3' raise."""

    def __init__(self, target_means=(0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1.), angle_range='v1',
                 add_ctr_clamp=False, ctr_clamp=32):
        self.means, self.stds = target_means, target_stds
        self.angle_range = angle_range
        self.add_ctr_clamp, self.ctr_clamp = add_ctr_clamp, ctr_clamp

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == 5 and gt_bboxes.size(-1) == 5
        if self.angle_range != 'v1':
            raise NotImplementedError(f'angle_range {self.angle_range!r}')
        return bbox2delta_v1(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        if self.angle_range != 'v1':
            raise NotImplementedError(f'angle_range {self.angle_range!r}')
        return delta2bbox_v1(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip,
                             self.add_ctr_clamp, self.ctr_clamp)
