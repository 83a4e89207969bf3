"""multiclass_nms_rotated -- mirror of r3det/core/post_processing/bbox_nms_rotated.py:7-131.

``nms`` is the config dict ``dict(type=..., iou_thr=...)``; ``type`` in {'v1' (default), 'v2',
'v3', 'mmcv'} picks the operator family exactly as the reference does (:43-125).
"""
import torch

from ... import _C
from ...ops import batched_rnms, ml_nms_rotated, obb_batched_nms
from ...ops.mmcv_ops import nms_rotated


MAX_CAP = 65472  # largest per-image candidate capacity of the batched kernels (pair queue packs (i << 16) | j)


class CapacityHint:
    """Caller-owned memory of the largest per-image candidate count seen for a shape.  With a hint,
    ``multiclass_nms_rotated_batch`` sizes its workspace from the previous call instead of reading the
    counts in the middle of the pipeline (one host synchronisation less per step); without one it always
    reads them.  One hint per caller (a head instance, a benchmark loop): nothing is shared behind the
    callers' backs; a lock makes concurrent use from several threads safe."""

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._last = {}

    def get(self, key):
        with self._lock:
            return self._last.get(key)

    def put(self, key, value):
        with self._lock:
            self._last[key] = value

    # A hint is a cache, not state: a copy (copy.deepcopy of the owning module for an EMA / SWA twin, torch.save of
    # the whole module, pickling into a worker process) starts empty with a lock of its own.
    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()

    def __deepcopy__(self, memo):
        return CapacityHint()


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


def multiclass_nms_rotated_batch(multi_bboxes, multi_scores, score_thr, nms, max_num=-1, hint=None):
    """``[multiclass_nms_rotated(b, s, ...) for b, s in zip(multi_bboxes, multi_scores)]`` for a
    whole batch -- (B, n, 5) boxes shared by the classes, (B, n, C+1) scores -- with identical
    results.  nms types 'v1', 'v2' and 'v3' run as ONE pass of launches over all images
    (r3det_mcnms_select / r3det_mcnms in include/r3det_hip.h: threshold + ordered compaction,
    stable score sort, class offsets or label guard, suppression, keep order and the max_num cut
    on the device; the host reads the per-image candidate counts in the middle -- or, when the caller
    passes its ``hint`` (CapacityHint) and the shape has been seen, together with the per-image detection
    counts at the end).  'mmcv', and pools with more than MAX_CAP candidates in one image, take the
    per-image path."""
    B, n = multi_scores.shape[:2]
    K = multi_scores.size(2) - 1
    version = nms.get('type', 'v1')
    iou_thr = float(_get(nms, 'iou_thr'))
    geom = {'v1': 1, 'v2': 2, 'v3': 3}.get(version, 0)
    fused = (geom and multi_bboxes.dim() == 3 and multi_bboxes.size(2) == 5 and n > 0 and K > 0
             and iou_thr >= 0)
    if not fused:
        return [multiclass_nms_rotated(multi_bboxes[i], multi_scores[i], score_thr, nms, max_num) for i in range(B)]
    boxes = _C.need_hip(multi_bboxes.contiguous(), "multi_bboxes")
    scores = _C.need_hip(multi_scores.contiguous(), "multi_scores")
    dev = boxes.device
    L = _C.lib()
    with torch.cuda.device(dev):
        S = n * K
        sel_bytes = int(L.r3det_mcnms_select_workspace_bytes(B, n))
        # the candidate arrays, the counters and the selection's workspace: ONE allocation carved into views (nine
        # torch.empty calls were ~25 us of host time per step, a tenth of the whole custom-op hot path)
        words = 4 * B * S + 2 * B + B
        pad = (-words) % 64
        blob = torch.empty(4 * (words + pad) + sel_bytes, dtype=torch.uint8, device=dev)
        i32 = blob[:4 * words].view(torch.int32)
        cand_row, cand_label, cand_rank = (i32[k * B * S:(k + 1) * B * S].view(B, S) for k in (0, 1, 2))
        cand_score = i32[3 * B * S:4 * B * S].view(torch.float32).view(B, S)
        kc = i32[4 * B * S:4 * B * S + 2 * B]  # [detections kept | candidate counts]: one host read
        kept_d, counts = kc[:B], kc[B:]
        maxc = i32[4 * B * S + 2 * B:words].view(torch.float32)
        sel_ws = blob[4 * (words + pad):]

        def buffers(cap):
            out_cap = max_num if max_num > 0 else cap
            ws_bytes = int(L.r3det_mcnms_workspace_bytes(B, cap))
            return (out_cap, ws_bytes, torch.empty(ws_bytes, dtype=torch.uint8, device=dev),
                    torch.empty((B, out_cap, 6), dtype=torch.float32, device=dev),
                    torch.empty((B, out_cap), dtype=torch.int64, device=dev))

        # cap (>= the largest candidate count) sizes the suppression workspace.  Reading the counts after
        # the selection costs a host synchronisation in the middle of the pipeline; once a shape has been
        # seen, cap is guessed from its last counts instead, everything is allocated up front, both library
        # calls go out back to back and the counts come back together with the results (the library clamps
        # an image to cap candidates, so a short guess is detected and redone, never unsafe).
        key = (dev, B, n, K)
        last = hint.get(key) if hint is not None else None
        guess = last is not None and last <= MAX_CAP
        if guess:
            cap = min(MAX_CAP, max(1024, (int(last * 1.3) + 63) // 64 * 64))
            out_cap, ws_bytes, ws, dets, labels = buffers(cap)
        _C.check(L.r3det_mcnms_select(_C.ptr(boxes), _C.ptr(scores), B, n, K, float(score_thr), _C.ptr(cand_row),
                                      _C.ptr(cand_label), _C.ptr(cand_score), _C.ptr(cand_rank), _C.ptr(counts),
                                      _C.ptr(maxc), _C.ptr(sel_ws), sel_bytes, _C.stream()), "r3det_mcnms_select")
        for _ in (0, 1):
            if not guess:
                m = int(counts.max().item())
                if hint is not None:
                    hint.put(key, m)
                if m == 0:
                    return [(multi_bboxes.new_zeros((0, 6)), multi_bboxes.new_zeros((0, ), dtype=torch.long))
                            for _ in range(B)]
                if m > MAX_CAP:  # beyond the pair-queue encoding (cap = ceil64(m) must stay < 65536)
                    return [multiclass_nms_rotated(multi_bboxes[i], multi_scores[i], score_thr, nms, max_num)
                            for i in range(B)]
                cap = (m + 63) // 64 * 64
                out_cap, ws_bytes, ws, dets, labels = buffers(cap)
            _C.check(L.r3det_mcnms(geom, _C.ptr(boxes), B, n, K, _C.ptr(cand_row), _C.ptr(cand_label),
                                   _C.ptr(cand_score), _C.ptr(cand_rank), _C.ptr(counts), _C.ptr(maxc), cap, iou_thr,
                                   out_cap, _C.ptr(ws), ws_bytes, _C.ptr(dets), _C.ptr(labels), None, _C.ptr(kept_d),
                                   _C.stream()), "r3det_mcnms")
            both = kc.tolist()
            kept, m = both[:B], max(both[B:])
            if hint is not None:
                hint.put(key, m)
            if m <= cap:
                break
            guess = False  # more candidates than guessed: once more with the exact size
        if geom == 2 and max_num <= 0:  # the reference's v2 branch slices [:max_num] whenever kept > max_num (:63-65)
            kept = [max(k + max_num, 0) if max_num < 0 else 0 for k in kept]
    return [(dets[i, :kept[i]], labels[i, :kept[i]]) for i in range(B)]


class PaddedNms:
    """``multiclass_nms_rotated`` for a whole batch with NO host synchronisation (round 5): the result is the padded
    tensor a detector hands on -- what ``rbbox2result`` (core/bbox/rtransforms.py:10-25, models/detectors/r3det.py:137-143)
    or the image-parallel gather consumes -- instead of per-image lists whose lengths the host has to read.

    ``out = nms(multi_bboxes (B, n, 5), multi_scores (B, n, C + 1))`` -> ``out`` (B, max_num + 1, 7) fp32: rows
    [0, max_num) of image i = [cx, cy, w, h, theta, score, label], zero beyond its count; row ``max_num`` =
    [count, 0, ...] (the layout ``dist_infer.gather_detections`` sends).  ``nms.counts`` (B,) int32 holds the counts,
    ``nms.overflow`` (B,) int32 is 1 where an image had more candidates than the fixed capacity: its rows are then the
    result for its first ``cap`` candidates.  Both calls of the pipeline (r3det_mcnms_select, r3det_mcnms_padded) are
    plain enqueues on the current stream with every buffer allocated up front, so the step can be recorded in a HIP
    graph.  The caller looks at ``overflow`` when it next touches the host anyway (``post_flags()`` behind a step: a
    pinned 4 * B byte copy; ``check(lag)`` one or two steps later; ``read()`` where the counts are read) and
    ``grow()``s + repeats that step in the rare case.

    Rows and values equal ``multiclass_nms_rotated_batch``'s lists (tests/test_gpu_mcnms.py)."""

    def __init__(self, B, n, K, score_thr, nms, max_num, cap, device):
        version = nms.get('type', 'v1')
        self.geom = {'v1': 1, 'v2': 2, 'v3': 3}.get(version, 0)
        if not self.geom or max_num <= 0:
            raise NotImplementedError("PaddedNms: nms types v1 / v2 / v3 with max_num > 0")
        self.B, self.n, self.K, self.max_num = B, n, K, int(max_num)
        self.score_thr, self.iou_thr = float(score_thr), float(_get(nms, 'iou_thr'))
        self.device = device
        self._alloc(cap)
        self._pending = []      # [(pinned flags, event)] of the steps whose flags were posted and not looked at yet
        self._free = []         # pinned buffers to use again

    def _alloc(self, cap):
        L = _C.lib()
        B, n, K, dev = self.B, self.n, self.K, self.device
        self.cap = min(MAX_CAP, max(64, (int(cap) + 63) // 64 * 64))
        S = n * K
        with torch.cuda.device(dev):
            self.sel_bytes = int(L.r3det_mcnms_select_workspace_bytes(B, n))
            self.ws_bytes = int(L.r3det_mcnms_workspace_bytes(B, self.cap))
            words = 4 * B * S + 4 * B
            pad = (-words) % 64
            sel_room = (self.sel_bytes + 255) // 256 * 256  # (the suppression workspace behind it stays 256-byte aligned)
            blob = torch.empty(4 * (words + pad) + sel_room + self.ws_bytes, dtype=torch.uint8, device=dev)
        i32 = blob[:4 * words].view(torch.int32)
        self.cand_row, self.cand_label, self.cand_rank = (i32[k * B * S:(k + 1) * B * S] for k in (0, 1, 2))
        self.cand_score = i32[3 * B * S:4 * B * S].view(torch.float32)
        self.counts = i32[4 * B * S:4 * B * S + B]        # detections kept
        self.overflow = i32[4 * B * S + B:4 * B * S + 2 * B]   # (next to the counts: lists() reads both in one copy)
        self.overflow.zero_()
        self._counts_flags = i32[4 * B * S:4 * B * S + 2 * B]
        self.cand_counts = i32[4 * B * S + 2 * B:4 * B * S + 3 * B]
        self.maxc = i32[4 * B * S + 3 * B:words].view(torch.float32)
        self.sel_ws = blob[4 * (words + pad):4 * (words + pad) + self.sel_bytes]
        self.ws = blob[4 * (words + pad) + sel_room:]
        self._blob = blob
        self.out = torch.zeros((B, self.max_num + 1, 7), dtype=torch.float32, device=dev)

    def grow(self, factor=2.0):
        """A larger candidate capacity (after ``check()`` reported an overflow).  New buffers: a graph that recorded
        the old ones has to be recorded again."""
        if self.cap >= MAX_CAP:
            raise RuntimeError("PaddedNms: more candidates per image than the batched kernels take; use the list form")
        self._alloc(int(self.cap * factor))

    def __call__(self, multi_bboxes, multi_scores):
        boxes = _C.need_hip(multi_bboxes, "multi_bboxes")
        scores = _C.need_hip(multi_scores, "multi_scores")
        B, n, K = self.B, self.n, self.K
        if tuple(boxes.shape) != (B, n, 5) or tuple(scores.shape) != (B, n, K + 1):
            raise RuntimeError(f"PaddedNms was built for boxes {(B, n, 5)} / scores {(B, n, K + 1)}")
        L = _C.lib()
        out = self.out
        _C.check(L.r3det_mcnms_select(_C.ptr(boxes), _C.ptr(scores), B, n, K, self.score_thr, _C.ptr(self.cand_row),
                                      _C.ptr(self.cand_label), _C.ptr(self.cand_score), _C.ptr(self.cand_rank),
                                      _C.ptr(self.cand_counts), _C.ptr(self.maxc), _C.ptr(self.sel_ws), self.sel_bytes,
                                      _C.stream()), "r3det_mcnms_select")
        rows = self.max_num + 1
        import ctypes
        count_row = ctypes.c_void_p(out.data_ptr() + 4 * self.max_num * 7)  # out[0, max_num, 0]
        _C.check(L.r3det_mcnms_padded(self.geom, _C.ptr(boxes), B, n, K, _C.ptr(self.cand_row), _C.ptr(self.cand_label),
                                      _C.ptr(self.cand_score), _C.ptr(self.cand_rank), _C.ptr(self.cand_counts),
                                      _C.ptr(self.maxc), self.cap, self.iou_thr, self.max_num, _C.ptr(self.ws),
                                      self.ws_bytes, _C.ptr(out), rows * 7, _C.ptr(self.counts), count_row, rows * 7,
                                      _C.ptr(self.overflow), _C.stream()), "r3det_mcnms_padded")
        return out

    def post_flags(self):
        """Behind a step (outside a graph): start the 4 * B byte copy of the overflow flags to pinned memory.  Every
        posted step keeps its own pinned buffer and event until ``check`` has looked at it."""
        host = self._free.pop() if self._free else torch.zeros(self.B, dtype=torch.int32).pin_memory()
        host.copy_(self.overflow, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((host, ev))

    def check(self, lag=1):
        """True when a posted step that is at least ``lag`` posts old overflowed the capacity.  lag = 1: every posted
        step is looked at (the host waits for the newest copy, i.e. for the step just enqueued); lag = 2: the newest
        post stays pending, so the host never waits for the step it enqueued last -- in a back-to-back loop the wait
        is for a step the GPU finished a whole step ago."""
        over = False
        while len(self._pending) >= max(1, lag):
            host, ev = self._pending.pop(0)
            ev.synchronize()
            over |= bool(host.any())
            self._free.append(host)
        return over

    def pending(self):
        """How many posted steps ``check`` has not looked at yet."""
        return len(self._pending)

    def read(self):
        """(kept counts, overflow flags) of the last call as Python lists: ONE device-to-host copy (a host
        synchronisation)."""
        both = self._counts_flags.tolist()
        return both[:self.B], both[self.B:]

    def lists(self):
        """The reference's per-image return values from the padded result (reads the counts: a host synchronisation;
        for tests and for callers that want ``multiclass_nms_rotated``'s lists)."""
        kept, _ = self.read()
        return [(self.out[i, :k, :6], self.out[i, :k, 6].long()) for i, k in enumerate(kept)]
