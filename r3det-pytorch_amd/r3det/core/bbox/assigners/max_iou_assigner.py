"""MaxIoUAssigner for rotated boxes -- the consumer of the (gt x anchors) overlaps on the training
path (models/dense_heads/rotate_anchor_head.py:220-231 builds it from
``dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1,
iou_calculator=dict(type='RBboxOverlaps2D_v1'))``, configs/r3det/r3det_r50_fpn_1x_dota_v1.py).

The class is third-party in the reference stack (mmdet 2.19
``mmdet/core/bbox/assigners/max_iou_assigner.py``, not under the reference tree), so its rules are
restated here: ``assign_wrt_overlaps`` is that restatement on a dense overlap matrix;
``assign`` runs the fused device path (``r3det_rbbox_assign``: no K x N matrix, SURVEY.md 8f
rank 3) whenever the configuration allows it and must return the same result
(tests/test_gpu_assign.py).  argmax ties resolve to the smaller index.
"""
import torch

from .... import _C
from ....registry import build_iou_calculator
from .. import iou_calculators  # noqa: F401  (registers RBboxOverlaps2D_v1/_v2/_v3)

_GEOM = {'RBboxOverlaps2D_v1': 1, 'RBboxOverlaps2D_v2': 2, 'RBboxOverlaps2D_v3': 3}


class AssignResult:
    """num_gts, gt_inds (0 = negative, -1 = ignore, i + 1 = gt i), max_overlaps, labels."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels

    @property
    def num_preds(self):
        return len(self.gt_inds)


_ASSIGN_PLANS = {}   # (n_gt, n_boxes) -> (workspace bytes, offsets of the result views in the call's one allocation)


class MaxIoUAssigner:
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1,
                 iou_calculator=dict(type='RBboxOverlaps2D_v1')):
        self.pos_iou_thr = pos_iou_thr
        self.neg_iou_thr = neg_iou_thr
        self.min_pos_iou = min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr = ignore_iof_thr
        self.ignore_wrt_candidates = ignore_wrt_candidates
        self.match_low_quality = match_low_quality
        self.gpu_assign_thr = gpu_assign_thr
        self.iou_calculator = build_iou_calculator(iou_calculator)

    # ------------------------------------------------------------------ fused device path
    def _fusable(self, bboxes, gt_bboxes, gt_bboxes_ignore):
        geom = _GEOM.get(type(self.iou_calculator).__name__)
        if geom is None or not isinstance(self.neg_iou_thr, float) or not bboxes.is_cuda:
            return None
        if self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0:
            return None
        if gt_bboxes.size(0) == 0 or bboxes.size(0) == 0:
            return None
        if geom == 3:
            # obb_overlaps zeroes rows / columns of boxes thinner than 1e-3 after its kernel
            if (gt_bboxes[:, 2:4].min() < 1e-3) or (bboxes[:, 2:4].min() < 1e-3):
                return None
        return geom

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None, with_gt_stats=False, shared_key=None):
        """``shared_key`` (not in mmdet's signature): a hashable that names ``bboxes`` when they are the SAME list in
        every call -- the anchor grid of a training run: what the kernels need of it is then computed once
        (r3det_iou_prepare_columns) and reused, whatever tensor object carries the boxes."""
        geom = self._fusable(bboxes, gt_bboxes, gt_bboxes_ignore)
        if geom is None:
            overlaps = self.iou_calculator(gt_bboxes, bboxes)
            # ignore regions (mmdet 2.19 max_iou_assigner.py, restated; off in the rotated configs: ignore_iof_thr = -1)
            if (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0
                    and bboxes.numel() > 0):
                if self.ignore_wrt_candidates:
                    ign = self.iou_calculator(bboxes, gt_bboxes_ignore, mode='iof').max(dim=1)[0]
                else:
                    ign = self.iou_calculator(gt_bboxes_ignore, bboxes, mode='iof').max(dim=0)[0]
                overlaps[:, ign > self.ignore_iof_thr] = -1
            return self.assign_wrt_overlaps(overlaps, gt_labels)
        b = bboxes if bboxes.size(1) == 5 else bboxes[:, :5]
        g = gt_bboxes if gt_bboxes.size(1) == 5 else gt_bboxes[:, :5]
        b = _C.need_hip(b if b.is_contiguous() and b.dtype == torch.float32 else b.contiguous().float(), "bboxes")
        g = _C.need_hip(g if g.is_contiguous() and g.dtype == torch.float32 else g.contiguous().float(), "gt_bboxes")
        n1, n2 = g.size(0), b.size(0)
        L = _C.lib()
        lab = None
        if gt_labels is not None:
            lab = gt_labels if gt_labels.dtype == torch.int64 and gt_labels.is_contiguous() else gt_labels.long().contiguous()
        with torch.cuda.device(b.device):
            # ONE allocation for the six results (the int64 ones first, then the floats) and one for the kernels' workspace
            # -- its own, so that a caller who keeps the result does not keep the workspace's ~200 MB alive -- where
            # there were six; sizes cached per (n1, n2)
            plan = _ASSIGN_PLANS.get((n1, n2))
            if plan is None:
                if len(_ASSIGN_PLANS) >= 64:
                    _ASSIGN_PLANS.clear()
                nbytes = int(L.r3det_rbbox_assign_workspace_bytes(n1, n2))
                o_lab = 8 * (2 * n2 + n1)
                o_f = o_lab + 8 * n2
                o_ws = o_f + 4 * (n2 + n1)
                plan = _ASSIGN_PLANS[(n1, n2)] = (nbytes, o_lab, o_f, o_ws)
            nbytes, o_lab, o_f, o_ws = plan
            blob = torch.empty(o_ws, dtype=torch.uint8, device=b.device)
            i64 = blob[:o_f].view(torch.int64)
            gt_inds, argmax, gargmax, labels = i64[:n2], i64[n2:2 * n2], i64[2 * n2:2 * n2 + n1], i64[2 * n2 + n1:]
            f32 = blob[o_f:o_f + 4 * (n2 + n1)].view(torch.float32)
            max_ov, gmax = f32[:n2], f32[n2:]
            ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
            prep = None
            if shared_key is not None:
                cache = self.__dict__.setdefault('_prepared_columns', {})
                k = (shared_key, geom, n2, b.device)
                prep = cache.get(k)
                if prep is None:
                    pb = int(L.r3det_iou_prepared_bytes(n2))
                    prep = torch.empty(pb, dtype=torch.uint8, device=b.device)
                    _C.check(L.r3det_iou_prepare_columns(geom, _C.ptr(b), n2, _C.ptr(prep), pb, _C.stream()),
                             "iou_prepare_columns")
                    cache.clear()  # (one anchor grid at a time)
                    cache[k] = prep
            # (the labels ride in the kernel that writes gt_inds: mmdet's four elementwise steps -- cat, cast, clamp,
            # index -- were four launches per call and half of the call's host time)
            _C.check(L.r3det_rbbox_assign_labeled(
                geom, _C.ptr(g), n1, _C.ptr(b), n2, _C.ptr(prep), float(self.pos_iou_thr), float(self.neg_iou_thr),
                float(self.min_pos_iou), int(self.match_low_quality), int(self.gt_max_assign_all), _C.ptr(gt_inds),
                _C.ptr(max_ov), _C.ptr(argmax), _C.ptr(gmax), _C.ptr(gargmax), _C.ptr(lab),
                _C.ptr(labels) if lab is not None else None, _C.ptr(ws), nbytes, _C.stream()), "r3det_rbbox_assign_labeled")
        res = AssignResult(n1, gt_inds, max_ov, labels if lab is not None else None)
        if with_gt_stats:
            res.argmax_overlaps, res.gt_max_overlaps, res.gt_argmax_overlaps = argmax, gmax, gargmax
        return res

    # ------------------------------------------------------------------ dense restatement
    @staticmethod
    def _labels(gt_inds, gt_labels):
        if gt_labels is None:
            return None
        # mmdet: labels = -1; labels[pos] = gt_labels[gt_inds[pos] - 1] with pos = nonzero(gt_inds > 0) -- the same values as
        # one table look-up, without nonzero's trip to the host (round 5: it was the one synchronisation of an assign call)
        table = torch.cat([gt_labels.new_full((1, ), -1), gt_labels]).to(gt_inds.dtype)
        return table[gt_inds.clamp(min=0)]

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
        gt_inds = overlaps.new_full((num_bboxes, ), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            max_overlaps = overlaps.new_zeros((num_bboxes, ))
            if num_gts == 0:
                gt_inds[:] = 0
            labels = None if gt_labels is None else overlaps.new_full((num_bboxes, ), -1, dtype=torch.long)
            return AssignResult(num_gts, gt_inds, max_overlaps, labels)
        # first maximum on ties (torch's CPU rule; the device reductions of the stack leave it open)
        max_overlaps = overlaps.max(dim=0)[0]
        argmax_overlaps = (overlaps == max_overlaps[None]).int().argmax(dim=0)
        gt_max_overlaps = overlaps.max(dim=1)[0]
        gt_argmax_overlaps = (overlaps == gt_max_overlaps[:, None]).int().argmax(dim=1)
        if isinstance(self.neg_iou_thr, float):
            gt_inds[(max_overlaps >= 0) & (max_overlaps < self.neg_iou_thr)] = 0
        elif isinstance(self.neg_iou_thr, tuple):
            assert len(self.neg_iou_thr) == 2
            gt_inds[(max_overlaps >= self.neg_iou_thr[0]) & (max_overlaps < self.neg_iou_thr[1])] = 0
        pos = max_overlaps >= self.pos_iou_thr
        gt_inds[pos] = argmax_overlaps[pos] + 1
        if self.match_low_quality:
            for i in range(num_gts):
                if gt_max_overlaps[i] >= self.min_pos_iou:
                    if self.gt_max_assign_all:
                        gt_inds[overlaps[i, :] == gt_max_overlaps[i]] = i + 1
                    else:
                        gt_inds[gt_argmax_overlaps[i]] = i + 1
        res = AssignResult(num_gts, gt_inds, max_overlaps, self._labels(gt_inds, gt_labels))
        res.argmax_overlaps, res.gt_max_overlaps, res.gt_argmax_overlaps = argmax_overlaps, gt_max_overlaps, \
            gt_argmax_overlaps
        return res
