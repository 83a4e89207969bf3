"""DeltaXYWHAOBBoxCoder, angle version v1 (core/bbox/coder/delta_xywha_rbbox_coder.py:104-211):
elementwise torch, means 0 / stds 1 in the shipped configs."""
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
                  wh_ratio_clip=16 / 1000):
    """rois (..., 5), deltas (..., 5) -> boxes (..., 5); dw/dh clamped to |log(wh_ratio_clip)|,
    centres clamped to the image when max_shape=(H, W) is given (:142-211)."""
    m, s = _const(deltas, means), _const(deltas, stds)
    d = deltas * s + m
    max_ratio = abs(math.log(wh_ratio_clip))
    dw = d[..., 2].clamp(min=-max_ratio, max=max_ratio)
    dh = d[..., 3].clamp(min=-max_ratio, max=max_ratio)
    gw = rois[..., 2] * dw.exp()
    gh = rois[..., 3] * dh.exp()
    gx = rois[..., 0] + rois[..., 2] * d[..., 0]
    gy = rois[..., 1] + rois[..., 3] * d[..., 1]
    ga = rois[..., 4] + d[..., 4]
    if max_shape is not None:
        gx = gx.clamp(min=0, max=max_shape[1] - 1)
        gy = gy.clamp(min=0, max=max_shape[0] - 1)
    return torch.stack([gx, gy, gw, gh, ga], dim=-1)


def bbox2delta_v1(proposals, gt, means=(0., 0., 0., 0., 0.), stds=(1., 1., 1., 1., 1.)):
    """Inverse of delta2bbox_v1 (:104-139)."""
    p, g = proposals.float(), gt.float()
    d = torch.stack([(g[..., 0] - p[..., 0]) / p[..., 2], (g[..., 1] - p[..., 1]) / p[..., 3],
                     torch.log(g[..., 2] / p[..., 2]), torch.log(g[..., 3] / p[..., 3]),
                     g[..., 4] - p[..., 4]], dim=-1)
    return (d - d.new_tensor(means)) / d.new_tensor(stds)
