"""multiclass_nms_rotated -- mirror of r3det/core/post_processing/bbox_nms_rotated.py:7-131.

``nms`` is the config dict ``dict(type=..., iou_thr=...)``; ``type`` in {'v1' (default), 'v2',
'v3', 'mmcv'} picks the operator family exactly as the reference does (:43-125).
"""
import torch

from ...ops import batched_rnms, ml_nms_rotated, nms_rotated, obb_batched_nms


def _get(nms, key):
    return nms[key] if isinstance(nms, dict) else getattr(nms, key)


def multiclass_nms_rotated(multi_bboxes, multi_scores, score_thr, nms, max_num=-1,
                           score_factors=None, return_inds=False):
    """(n, 5 | C*5) boxes + (n, C+1) scores -> (dets (k,6), labels (k,)[, inds])."""
    num_classes = multi_scores.size(1) - 1
    n = multi_scores.size(0)
    if multi_bboxes.shape[1] > 5:
        bboxes = multi_bboxes.view(n, -1, 5)
    else:
        bboxes = multi_bboxes[:, None].expand(n, num_classes, 5)
    scores = multi_scores[:, :-1]  # last column = background
    version = nms.get('type', 'v1')
    iou_thr = _get(nms, 'iou_thr')

    if version == 'mmcv':
        labels = torch.arange(num_classes, dtype=torch.long, device=scores.device)
        labels = labels.view(1, -1).expand_as(scores).reshape(-1)
        bboxes = bboxes.reshape(-1, 5)
        scores = scores.reshape(-1)
        valid = scores > score_thr
        if score_factors is not None:
            scores = scores * score_factors.view(-1, 1).expand(n, num_classes).reshape(-1)
        inds = valid.nonzero(as_tuple=False).squeeze(1)
        bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
        if bboxes.numel() == 0:
            dets = torch.cat([bboxes, scores[:, None]], -1)
            return (dets, labels, inds) if return_inds else (dets, labels)
        dets, keep = nms_rotated(bboxes, scores, iou_thr, labels)
        if max_num > 0:
            dets, keep = dets[:max_num], keep[:max_num]
        return (dets, labels[keep], keep) if return_inds else (dets, labels[keep])

    valid = scores > score_thr
    bboxes = bboxes[valid]
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    scores = scores[valid]
    labels = valid.nonzero(as_tuple=False)[:, 1]  # row-major: anchor-major, class-minor
    if bboxes.numel() == 0:
        return multi_bboxes.new_zeros((0, 6)), multi_bboxes.new_zeros((0, ), dtype=torch.long)

    if version == 'v1':
        dets, keep = batched_rnms(bboxes, scores, labels, iou_thr)
    elif version == 'v3':
        dets, keep = obb_batched_nms(bboxes, scores, labels, iou_thr)
    elif version == 'v2':
        keep = ml_nms_rotated(bboxes, scores, labels, iou_thr)
        bboxes, scores, labels = bboxes[keep], scores[keep], labels[keep]
        if keep.size(0) > max_num:
            top = scores.sort(descending=True)[1][:max_num]
            bboxes, scores, labels = bboxes[top], scores[top], labels[top]
        return torch.cat([bboxes, scores[:, None]], 1), labels
    else:
        raise KeyError(f'unknown rotated nms type {version!r}')

    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, labels[keep]
