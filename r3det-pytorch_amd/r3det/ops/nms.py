"""Rotated NMS operators.

Host-side mirror of
  r3det/ops/rnms/rnms_wrapper.py:7-69                  rnms, batched_rnms       (v1)
  r3det/ops/nms_rotated/nms_rotated_wrapper.py:7-98    obb_nms, obb_batched_nms (v3)
  r3det/ops/ml_nms_rotated/__init__.py:1               ml_nms_rotated           (v2)
  mmcv.ops.nms_rotated (call site bbox_nms_rotated.py:86)                       (mmcv)
over include/r3det_hip.h.  Torch supplies the score sort, memory and the stream; mask build,
greedy reduction and (v1) the ascending re-sort run in libr3det_hip.so.  The only host
synchronisation is reading the 4-byte keep count.
"""
import numpy as np
import torch

from .. import _C


def _order(scores):
    # stable => ties keep the lower index first, like the reference's CPU sort
    return torch.sort(scores, descending=True, stable=True)[1]


def _run(fn_name, n, device, call):
    """Allocate workspace / outputs, run ``call(ws, ws_bytes, keep, count, stream)``."""
    L = _C.lib()
    with torch.cuda.device(device):
        ws_bytes = int(L.r3det_nms_workspace_bytes(n))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        keep = torch.empty(max(n, 1), dtype=torch.int64, device=device)
        count = torch.empty(1, dtype=torch.int32, device=device)
        _C.check(call(L, _C.ptr(ws), ws_bytes, _C.ptr(keep), _C.ptr(count), _C.stream()), fn_name)
        k = int(count.item())
    return keep[:k]


def rnms_ext_rnms(dets, thr):
    """rnms_ext.rnms(dets (n,6), thr) -> int64 keep, ascending (rnms_ext.cpp:11-19)."""
    dets = _C.need_hip(dets, "dets")
    n = dets.size(0)
    if n == 0:
        # the CUDA path returns a CPU tensor for empty input (rnms_cuda.cpp:9-11)
        return torch.empty(0, dtype=torch.long, device='cpu')
    if dets.dim() != 2 or dets.size(1) != 6:
        raise RuntimeError(f"dets must have shape (n, 6), got {tuple(dets.shape)}")
    order = _order(dets[:, 5])
    return _run("rnms", n, dets.device,
                lambda L, ws, wsb, keep, cnt, st: L.r3det_rnms(_C.ptr(dets), _C.ptr(order), n, float(thr), 1,
                                                               ws, wsb, keep, cnt, st))


def nms_rotated_ext_nms_rotated(dets, scores, thr):
    """nms_rotated_ext.nms_rotated(dets (n,5), scores, thr) -> keep, score order
    (nms_rotated_ext.cpp:24-35)."""
    dets = _C.need_hip(dets.contiguous(), "dets")
    scores = _C.need_hip(scores.contiguous(), "scores")
    assert dets.device == scores.device
    n = dets.size(0)
    if n == 0:
        return torch.empty(0, dtype=torch.long, device=dets.device)
    order = _order(scores)
    return _run("nms_rotated", n, dets.device,
                lambda L, ws, wsb, keep, cnt, st: L.r3det_nms_rotated(_C.ptr(dets), _C.ptr(order), n,
                                                                      float(thr), ws, wsb, keep, cnt, st))


def ml_nms_rotated(dets, scores, labels, iou_threshold):
    """ml_nms_rotated(dets (n,5), scores (n,), labels (n,), thr) -> keep in score order
    (ml_nms_rotated/src/nms_rotated.h:23-39).  Boxes with different labels never suppress
    each other."""
    dets = _C.need_hip(dets.contiguous(), "dets")
    scores = _C.need_hip(scores.contiguous(), "scores")
    if not labels.is_cuda:
        raise RuntimeError("labels must be a CUDA tensor")
    labels = labels.contiguous().to(torch.int64)
    n = dets.size(0)
    if n == 0:
        return torch.empty(0, dtype=torch.long, device=dets.device)
    order = _order(scores)
    return _run("ml_nms_rotated", n, dets.device,
                lambda L, ws, wsb, keep, cnt, st: L.r3det_ml_nms_rotated(
                    _C.ptr(dets), _C.ptr(labels), _C.ptr(order), n, float(iou_threshold), ws, wsb, keep,
                    cnt, st))


# mmcv 1.3.15..1.5.0's rotated NMS kernel copies the label column (stride-6 rows when ``labels`` is given)
# but its pair test ``single_box_iou_rotated(cur_box, block_boxes + i * 6, 0)`` reads the five box
# values only: with or without labels the suppression is CLASS-AGNOSTIC.  Restated from memory of the
# mmcv sources (they are not under the reference tree; parity UNPINNED).  True = use the label
# guard of the in-tree ml_nms_rotated instead (boxes of different labels never suppress each other).
# This constant is only the DEFAULT of ``nms_rotated(..., label_guard=None)``; pass the keyword to choose per call.
MMCV_LABEL_GUARD = False


def nms_rotated(dets, scores, iou_threshold, labels=None, label_guard=None):
    """Stand-in for ``mmcv.ops.nms_rotated``: returns (cat(dets[keep], scores[keep]), keep), keep in
    score order.

    mmcv is outside the reference tree (pinned only as 1.3.15..1.5.0); semantics restated from
    its call site bbox_nms_rotated.py:86-95, the in-tree ml_nms_rotated sources (geometry) and the
    note on ``MMCV_LABEL_GUARD`` above: labels are accepted and, by default, have no effect -- UNVERIFIED
    against a real mmcv build.  ``label_guard=True`` (not an mmcv keyword) makes boxes of different
    labels never suppress each other, as the in-tree ml_nms_rotated does; ``None`` takes the module default.
    """
    if dets.shape[0] == 0:
        return dets, None
    dets_c = _C.need_hip(dets.contiguous(), "dets")
    scores_c = _C.need_hip(scores.contiguous(), "scores")
    lab = None
    if labels is not None and (MMCV_LABEL_GUARD if label_guard is None else label_guard):
        lab = labels.to(device=dets.device, dtype=torch.int64).contiguous()
    n = dets_c.size(0)
    order = _order(scores_c)
    keep = _run("mmcv_nms_rotated", n, dets.device,
                lambda L, ws, wsb, kp, cnt, st: L.r3det_mmcv_nms_rotated(
                    _C.ptr(dets_c), _C.ptr(lab) if lab is not None else None, _C.ptr(order), n,
                    float(iou_threshold), ws, wsb, kp, cnt, st))
    out = torch.cat((dets[keep], scores[keep].reshape(-1, 1)), dim=1)
    return out, keep


def _numpy_in(dets, device_id):
    if isinstance(dets, torch.Tensor):
        return False, dets
    if isinstance(dets, np.ndarray):
        dev = torch.device('cuda', torch.cuda.current_device() if device_id is None else device_id)
        return True, torch.from_numpy(dets).to(dev)
    raise TypeError(f'dets must be either a Tensor or numpy array, but got {type(dets)}')


def rnms(dets, nms_thr, device_id=None):
    """NMS v1 on (n, 6) ``[cx, cy, w, h, theta, score]`` -> (dets[keep], keep), keep ascending
    (rnms_wrapper.py:7-31).  numpy input is staged on ``cuda:device_id`` (no CPU path here)."""
    is_numpy, d = _numpy_in(dets, device_id)
    if d.shape[0] == 0:
        inds = d.new_zeros(0, dtype=torch.long)
    else:
        inds = rnms_ext_rnms(d.contiguous().float() if d.dtype != torch.float32 else d.contiguous(),
                             nms_thr)
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds


def batched_rnms(bboxes, scores, inds, nms_thr, class_agnostic=False):
    """Per-class NMS v1 through coordinate offsets (rnms_wrapper.py:34-69).

    Offset = label * (max over ALL five box columns + 1), added to cx, cy.  Returns
    (cat(bboxes[keep], scores[keep]), keep) with keep ascending.
    """
    fast = _batched_rnms_device(bboxes, scores, inds, nms_thr, class_agnostic)
    if fast is not None:
        return fast
    if class_agnostic:
        shifted = bboxes
    else:
        offs = inds.to(bboxes) * (bboxes.max() + 1)
        shifted = bboxes.clone()
        shifted[:, :2] += offs[:, None]
    dets, keep = rnms(torch.cat([shifted, scores[:, None]], -1), nms_thr)
    return torch.cat([bboxes[keep], dets[:, -1:]], -1), keep


_RNMS_WS = {}
# Largest pool the one-call forms take.  The library holds up to 65 472 rows, but its workspace carries a dense
# n x n/64 suppressor mask (540 MB, zero-filled per call, at 65 k rows).  Measured in round 6 (us per call, op-by-op
# wrapper / one call): 16 000: 163 / 163, 24 576: 398 / 362, 32 768: 611 / 525 (then 4 candidates per thread in the
# ranking loop), 49 152: 1035 / 1078, 65 000: 1997 / 2169 -- the crossover lies between 32 768 and 49 152.
FAST_MAX_N = 32768


def _batched_rnms_device(bboxes, scores, inds, nms_thr, class_agnostic, entry="r3det_batched_rnms", padded=False):
    """The same result from ONE library call (r3det_batched_rnms: candidate arrays and the wrapper's bboxes.max()
    from one small kernel, stable score sort by counting, class offsets, suppression, ascending keep and the gather
    on the device) instead of arange / zeros / max / mul / clone / add / cat / sort / rnms / index launches.  None
    when the input does not qualify (then the op-by-op form above runs)."""
    if not (isinstance(bboxes, torch.Tensor) and bboxes.is_cuda and bboxes.dtype == torch.float32
            and bboxes.dim() == 2 and bboxes.size(1) == 5 and isinstance(scores, torch.Tensor)
            and scores.dtype == torch.float32 and scores.device == bboxes.device
            and (class_agnostic or (isinstance(inds, torch.Tensor) and inds.device == bboxes.device
                                    and not inds.dtype.is_floating_point and inds.dtype != torch.bool))):
        return None  # (float ``inds`` multiply the offset as floats in the reference: op-by-op form)
    n = bboxes.size(0)
    if n == 0 or n > min(FAST_MAX_N, 65472) or not (nms_thr >= 0):  # (65 472 = the library's row capacity)
        return None
    dev = bboxes.device
    L = _C.lib()
    with torch.cuda.device(dev):
        b = bboxes.contiguous()
        sc = scores.contiguous()
        lab = None if class_agnostic else inds.to(torch.int64).contiguous()
        ws_bytes = _RNMS_WS.get(n)
        if ws_bytes is None:
            ws_bytes = _RNMS_WS[n] = int(L.r3det_batched_rnms_workspace_bytes(n))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        dets = torch.empty((n, 6), dtype=torch.float32, device=dev)
        keep = torch.empty(n, dtype=torch.int64, device=dev)
        kept = torch.empty(1, dtype=torch.int32, device=dev)
        _C.check(getattr(L, entry)(_C.ptr(b), _C.ptr(sc), _C.ptr(lab) if lab is not None else None, n,
                                   float(nms_thr), _C.ptr(ws), ws_bytes, _C.ptr(dets), _C.ptr(keep), _C.ptr(kept),
                                   _C.stream()), entry)
        if padded:  # (no host read: the caller slices by ``kept`` when it next touches the host)
            return dets, keep, kept
        k = int(kept.item())
    return dets[:k], keep[:k]


def batched_rnms_padded(bboxes, scores, inds, nms_thr, class_agnostic=False, version='v1'):
    """``batched_rnms`` / ``obb_batched_nms`` (rnms_wrapper.py:34-69, nms_rotated_wrapper.py:78-98) WITHOUT the host read of
    the count: -> ``(dets (n, 6), keep (n,), kept (1,) int32)`` -- rows ``[:kept]`` of ``dets`` / ``keep`` are the
    reference's return values, the rest is unspecified.  The one library call of the list form, enqueued and returned:
    for pipelines that consume the result on the device (the padded multiclass form, a HIP graph).  None when the input
    does not qualify for the one-call form."""
    entry = {"v1": "r3det_batched_rnms", "v3": "r3det_obb_batched_nms"}[version]
    return _batched_rnms_device(bboxes, scores, inds, nms_thr, class_agnostic, entry=entry, padded=True)


def obb2hbb(obboxes):
    """Circumscribed horizontal boxes [x1, y1, x2, y2] (nms_rotated_wrapper.py:7-20)."""
    ctr, w, h, theta = torch.split(obboxes, [2, 1, 1, 1], dim=1)
    c, s = torch.cos(theta), torch.sin(theta)
    half = torch.cat([torch.abs(w / 2 * c) + torch.abs(h / 2 * s),
                      torch.abs(w / 2 * s) + torch.abs(h / 2 * c)], dim=1)
    return torch.cat([ctr - half, ctr + half], dim=1)


def obb_nms(dets, iou_thr, device_id=None):
    """NMS v3 on (n, 6) dets -> (dets[keep], keep) in score order (nms_rotated_wrapper.py:23-53).
    Boxes with min(w, h) < 1e-3 are removed before the kernel and never kept."""
    is_numpy, d = _numpy_in(dets, device_id)
    if d.numel() == 0:
        inds = d.new_zeros(0, dtype=torch.int64)
    else:
        ok = ~(d[:, 2:4].min(1)[0] < 0.001)
        ori = torch.nonzero(ok, as_tuple=False).squeeze(1)
        if ori.numel() == 0:
            inds = d.new_zeros(0, dtype=torch.int64)
        else:
            dd = d[ori]
            inds = ori[nms_rotated_ext_nms_rotated(dd[:, :5], dd[:, 5], iou_thr)]
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds


def obb_batched_nms(bboxes, scores, inds, nms_thr, class_agnostic=False):
    """Per-class NMS v3 (nms_rotated_wrapper.py:78-98); offset = label * (hbb extent + 1)."""
    if isinstance(bboxes, torch.Tensor) and bboxes.dim() == 2 and bboxes.size(-1) == 5:
        fast = _batched_rnms_device(bboxes, scores, inds, nms_thr, class_agnostic, "r3det_obb_batched_nms")
        if fast is not None:  # (one library call; the op-by-op form below is ~10 launches and a nonzero() sync)
            return fast
    if class_agnostic:
        shifted = bboxes
    else:
        hbb = obb2hbb(bboxes) if bboxes.size(-1) == 5 else bboxes
        offs = inds.to(bboxes) * (hbb.max() - hbb.min() + 1)
        if bboxes.size(-1) == 5:
            shifted = bboxes.clone()
            shifted[:, :2] = shifted[:, :2] + offs[:, None]
        else:
            shifted = bboxes + offs[:, None]
    dets, keep = obb_nms(torch.cat([shifted, scores[:, None]], -1), nms_thr)
    return torch.cat([bboxes[keep], dets[:, -1:]], -1), keep


def nms_rotated_ext_nms_poly(dets9, thr):
    """nms_rotated_ext.nms_poly(dets (n,9), thr) -> keep in score order (nms_rotated_ext.cpp:38-51;
    empty input -> CPU int64(0) like the reference)."""
    dets9 = _C.need_hip(dets9.contiguous(), "dets")
    n = dets9.size(0)
    if n == 0:
        return torch.empty(0, dtype=torch.long, device='cpu')
    if dets9.dim() != 2 or dets9.size(1) != 9:
        raise RuntimeError(f"dets must have shape (n, 9), got {tuple(dets9.shape)}")
    order = _order(dets9[:, 8])
    L = _C.lib()
    with torch.cuda.device(dets9.device):
        ws_bytes = int(L.r3det_poly_nms_workspace_bytes(n))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dets9.device)
        keep = torch.empty(n, dtype=torch.int64, device=dets9.device)
        count = torch.empty(1, dtype=torch.int32, device=dets9.device)
        _C.check(L.r3det_nms_poly(_C.ptr(dets9), _C.ptr(order), n, float(thr), _C.ptr(ws), ws_bytes, _C.ptr(keep),
                                  _C.ptr(count), _C.stream()), "r3det_nms_poly")
        k = int(count.item())
    return keep[:k]


def poly_nms(dets, iou_thr, device_id=None):
    """NMS of 8-coordinate polygons, dets (n, 9) = 8 coordinates + score -> (dets[keep], keep) in
    score order (nms_rotated_wrapper.py:56-75; poly_nms_cuda.cu).  Used by the DOTA result merge
    (datasets/dota1.py:654).  Like the reference, CPU tensors are refused (NotImplementedError);
    numpy input is staged on ``cuda:device_id``."""
    if isinstance(dets, torch.Tensor):
        is_numpy, d = False, dets
    elif isinstance(dets, np.ndarray):
        is_numpy = True
        if device_id is None:
            raise NotImplementedError  # the reference maps device_id=None to 'cpu', which it refuses
        d = torch.from_numpy(dets).to(torch.device('cuda', device_id))
    else:
        raise TypeError(f'dets must be eithr a Tensor or numpy array, but got {type(dets)}')
    if d.device.type == 'cpu':
        raise NotImplementedError
    inds = nms_rotated_ext_nms_poly(d.float(), iou_thr).to(d.device)
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds
