"""The two losses the r3det configs select (configs/r3det/r3det_r50_fpn_1x_dota_v1.py:37-43,62-68):
``FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25)`` and ``SmoothL1Loss(beta=0.11)``.

Both classes are mmdet 2.19's (third-party, not under the reference tree); their rules are restated:
elementwise loss, times the per-sample weight, reduced as ``sum / avg_factor`` when ``avg_factor`` is
given with ``reduction='mean'`` (mmdet ``weight_reduce_loss``), times ``loss_weight``.  ``avg_factor``
may be a tensor (the positive count stays on the device: no host synchronisation in the step).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return {'none': loss, 'mean': loss.mean(), 'sum': loss.sum()}[reduction]
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    """pred (n, C) logits, target (n,) class index with C = background.  Per element
    ``-alpha (1 - p)^gamma log p`` for the target class, ``-(1 - alpha) p^gamma log(1 - p)`` otherwise
    (the formula of mmcv's sigmoid_focal_loss op that mmdet calls on the device)."""
    C = pred.size(1)
    t = F.one_hot(target, num_classes=C + 1)[:, :C].to(pred.dtype)
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    focal = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * focal
    if weight is not None and weight.shape != loss.shape:
        weight = weight.view(-1, 1) if weight.size(0) == loss.size(0) else weight.view(loss.size(0), -1)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid, 'only the sigmoid focal loss is used by the rotated heads'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override or self.reduction
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, self.gamma, self.alpha, reduction,
                                                     avg_factor)


def smooth_l1_loss(pred, target, beta=1.0):
    assert beta > 0
    if target.numel() == 0:
        return pred.sum() * 0
    diff = (pred - target).abs()
    return torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)


class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override or self.reduction
        loss = smooth_l1_loss(pred, target, self.beta)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


class L1Loss(nn.Module):
    """mmdet L1Loss (configs/rretinanet/rretinanet_obb_r50_fpn_1x_dota_v1.py:49)."""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override or self.reduction
        return self.loss_weight * weight_reduce_loss((pred - target).abs(), weight, reduction, avg_factor)


from ..registry import LOSSES, build_from  # noqa: E402

for _c in (FocalLoss, SmoothL1Loss, L1Loss):
    if _c.__name__ not in LOSSES:
        LOSSES.register_module(module=_c)


def build_loss(cfg):
    return build_from(LOSSES, cfg)
