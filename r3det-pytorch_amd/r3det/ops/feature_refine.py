"""Feature Refinement Module (rotated feature-align sampler).

Host-side mirror of r3det/ops/fr/feature_refine_module.py:13-127 over
r3det_feature_refine_forward / _backward (include/r3det_hip.h).  Module, parameter and
state-dict names (``conv_5_1``, ``conv_1_5``, ``conv_1_1``, ``fr``) are the reference's.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C


# bench.py hook: when set to a list, (start, end) stream events are recorded around every
# forward launch whose plane is >= PROFILE_MIN_HW (the level-0 launch of a 1024^2 input)
profile_events = None
PROFILE_MIN_HW = 128 * 128


def _taken(rc, what):
    """Split-form entry points answer R3DET_EINVAL (-1) for shapes they do not take (nothing was launched:
    the caller uses the general form); any other non-zero code is a real failure and raises."""
    if rc == 0:
        return True
    if rc == -1:
        return False
    _C.check(rc, what)


def fr_forward(features, best_rbboxes, spatial_scale, points, output):
    """feature_refine_cuda.forward (feature_refine_cuda.cpp:24-42): fills ``output``, returns 1."""
    f = _C.need_hip(features, "features")
    b = _C.need_hip(best_rbboxes, "best_bboxes")
    o = _C.need_hip(output, "output")
    N, C, H, W = f.shape
    if b.numel() != N * H * W * 5:
        raise RuntimeError(f"best_bboxes must hold N*H*W x 5 values, got {tuple(b.shape)}")
    with torch.cuda.device(f.device):
        ev = None
        if profile_events is not None and H * W >= PROFILE_MIN_HW:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        L = _C.lib()
        wsb = int(L.r3det_fr_workspace_bytes(N, H, W, int(points)))
        ws = torch.empty(wsb, dtype=torch.uint8, device=f.device)
        _C.check(L.r3det_feature_refine_forward(_C.ptr(f), _C.ptr(b), N, C, H, W, float(spatial_scale),
                                                int(points), _C.ptr(o), _C.ptr(ws), wsb, _C.stream()), "fr_forward")
        if ev is not None:
            ev[1].record()
            profile_events.append(ev)
    return 1


def fr_forward_levels(features, best_rbboxes, spatial_scales, points, outputs):
    """The samplers of all pyramid levels in one library call (r3det_feature_refine_forward_levels):
    lists over levels of (N,C,H,W) features / outputs and (N*H*W, 5) boxes."""
    import ctypes
    n = len(features)
    fs = [_C.need_hip(f, "features") for f in features]
    bs = [_C.need_hip(b, "best_bboxes") for b in best_rbboxes]
    os_ = [_C.need_hip(o, "output") for o in outputs]
    N, C = fs[0].shape[:2]
    for f, b in zip(fs, bs):
        if f.size(0) != N or f.size(1) != C or b.numel() != N * f.size(2) * f.size(3) * 5:
            raise RuntimeError("levels must share N and C and bring N*H*W x 5 boxes each")
    arr_p = ctypes.c_void_p * n
    arr_i = ctypes.c_int * n
    H, W = arr_i(*[f.size(2) for f in fs]), arr_i(*[f.size(3) for f in fs])
    sc = (ctypes.c_float * n)(*[float(s) for s in spatial_scales])
    L = _C.lib()
    with torch.cuda.device(fs[0].device):
        wsb = int(L.r3det_fr_levels_workspace_bytes(n, N, H, W, int(points)))
        ws = torch.empty(wsb, dtype=torch.uint8, device=fs[0].device)
        _C.check(L.r3det_feature_refine_forward_levels(
            n, arr_p(*[f.data_ptr() for f in fs]), arr_p(*[b.data_ptr() for b in bs]), N, C, H, W, sc, int(points),
            arr_p(*[o.data_ptr() for o in os_]), _C.ptr(ws), wsb, _C.stream()), "fr_forward_levels")
    return 1


def fr_prepare(best_rbboxes, N, H, W, spatial_scale, points=1):
    """Tap table of one level for ``fr_forward_prepared`` (r3det_feature_refine_prepare), or None
    when the level has no split form (then use ``fr_forward``)."""
    if points != 1 or not best_rbboxes.is_cuda:
        return None
    L = _C.lib()
    nbytes = int(L.r3det_fr_table_bytes(N, H, W))
    if nbytes == 0:
        return None
    b = _C.need_hip(best_rbboxes.contiguous(), "best_bboxes")
    with torch.cuda.device(b.device):
        table = torch.empty(nbytes // 4, dtype=torch.float32, device=b.device)
        _C.check(L.r3det_feature_refine_prepare(_C.ptr(b), N, H, W, float(spatial_scale), _C.ptr(table),
                                                _C.stream()), "fr_prepare")
    return table


def fr_forward_prepared(features, table, output):
    """The sampler launch alone (r3det_feature_refine_forward_prepared).  Returns False when the
    library does not take this shape in the split form (nothing was launched)."""
    f = _C.need_hip(features, "features")
    o = _C.need_hip(output, "output")
    N, C, H, W = f.shape
    with torch.cuda.device(f.device):
        rc = _C.lib().r3det_feature_refine_forward_prepared(_C.ptr(f), _C.ptr(table), N, C, H, W, _C.ptr(o),
                                                            _C.stream())
    return _taken(rc, "fr_forward_prepared")


def fr_module_prepared(mixed_a, mixed_b, residual, table, output):
    """``residual + fr(mixed_a + mixed_b)`` in one launch (r3det_feature_refine_module_prepared): the
    module's two elementwise passes folded into the sampler.  False when the library does not take this
    shape (nothing was launched)."""
    a = _C.need_hip(mixed_a, "mixed_a")
    b = _C.need_hip(mixed_b, "mixed_b") if mixed_b is not None else None  # None: mixed_a is the summed plane
    r = _C.need_hip(residual, "residual")
    o = _C.need_hip(output, "output")
    N, C, H, W = a.shape
    if (b is not None and b.shape != a.shape) or r.shape != a.shape or o.shape != a.shape:
        raise RuntimeError("mixed_a, mixed_b, residual and output must have one shape")
    with torch.cuda.device(a.device):
        rc = _C.lib().r3det_feature_refine_module_prepared(_C.ptr(a), _C.ptr(b), _C.ptr(r), _C.ptr(table), N, C, H,
                                                           W, _C.ptr(o), _C.stream())
    return _taken(rc, "fr_module_prepared")


def _is_cl(t):
    """fp32 HIP tensor in channels_last memory that the NHWC kernels take (C % 4 == 0; not also NCHW-contiguous)."""
    return (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4
            and t.size(1) % 4 == 0 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last))


def _need_cl(t, name):
    """A 4-d fp32 HIP tensor whose memory is (N, H, W, C) contiguous (torch channels_last)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        raise RuntimeError(f"{name} must be a 4-d fp32 CUDA tensor")
    if not t.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError(f"{name} must be channels_last contiguous")
    return t


def fr_forward_nhwc(features, best_rbboxes, spatial_scale, points, output):
    """The sampler on channels_last memory (r3det_feature_refine_forward_nhwc): ``features`` / ``output`` are
    (N, C, H, W) tensors in torch.channels_last.  Same values as ``fr_forward``.  False when the library does
    not take the shape (C % 4 != 0): nothing was launched."""
    f, o = _need_cl(features, "features"), _need_cl(output, "output")
    b = _C.need_hip(best_rbboxes, "best_bboxes")
    N, C, H, W = f.shape
    if b.numel() != N * H * W * 5 or o.shape != f.shape:
        raise RuntimeError("best_bboxes must hold N*H*W x 5 values and output must have the features' shape")
    with torch.cuda.device(f.device):
        rc = _C.lib().r3det_feature_refine_forward_nhwc(_C.ptr(f), _C.ptr(b), N, C, H, W, float(spatial_scale),
                                                        int(points), _C.ptr(o), _C.stream())
    return _taken(rc, "fr_forward_nhwc")


def fr_module_nhwc(conv_a, conv_b, bias_a, bias_b, residual, best_rbboxes, spatial_scale, points, output):
    """The FeatureRefineModule tail for channels_last pipelines in ONE launch
    (r3det_feature_refine_module_nhwc): ``P = (conv_a + bias_a) + (conv_b + bias_b)``,
    ``output = residual + (P + sample(P))`` with conv_a / conv_b the raw (bias-free) outputs of
    conv_5_1(conv_1_5(x)) and conv_1_1(x).  All maps channels_last.  False when the shape is not taken."""
    a, r, o = _need_cl(conv_a, "conv_a"), _need_cl(residual, "residual"), _need_cl(output, "output")
    cb = _need_cl(conv_b, "conv_b") if conv_b is not None else None
    b = _C.need_hip(best_rbboxes, "best_bboxes")
    N, C, H, W = a.shape
    if b.numel() != N * H * W * 5 or r.shape != a.shape or o.shape != a.shape or (cb is not None and cb.shape != a.shape):
        raise RuntimeError("conv_a, conv_b, residual and output must have one shape; best_bboxes N*H*W x 5")
    for t, nm in ((bias_a, "bias_a"), (bias_b, "bias_b")):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or t.numel() != C or not t.is_contiguous()):
            raise RuntimeError(f"{nm} must be a contiguous fp32 CUDA tensor of C values")
    with torch.cuda.device(a.device):
        rc = _C.lib().r3det_feature_refine_module_nhwc(_C.ptr(a), _C.ptr(cb), _C.ptr(bias_a), _C.ptr(bias_b), _C.ptr(r),
                                                       _C.ptr(b), N, C, H, W, float(spatial_scale), int(points),
                                                       _C.ptr(o), _C.stream())
    return _taken(rc, "fr_module_nhwc")


def _lvl_arrays(feats, scales):
    import ctypes
    n = len(feats)
    arr_i = ctypes.c_int * n
    return (arr_i(*[f.size(2) for f in feats]), arr_i(*[f.size(3) for f in feats]),
            (ctypes.c_float * n)(*[float(s) for s in scales]), ctypes.c_void_p * n)


def _ptr_array(ts):
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])


def tap_tables(N, shapes, device):
    """One tap table per level ((H, W) in ``shapes``) carved from ONE allocation: per level ``N * 2 * H * W`` floats
    (r3det_fr_tap_table_bytes), 256-byte aligned.  The channels_last forward calls write them (``tables=``), the
    backward's index calls read them instead of the box records."""
    sizes = [((N * 2 * h * w + 63) // 64) * 64 for h, w in shapes]
    blob = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    out, at = [], 0
    for sz, (h, w) in zip(sizes, shapes):
        out.append(blob[at:at + N * 2 * h * w])
        at += sz
    return out


def fr_forward_levels_nhwc(features, best_rbboxes, spatial_scales, points, outputs, tables=None):
    """The channels_last sampler of all pyramid levels in one library call
    (r3det_feature_refine_forward_levels_nhwc): the level that takes the wide form alone, all others as ONE launch.
    ``tables`` (points = 1): per level a tap table (``tap_tables``) the launches also write, or None.
    False when nothing was launched (argument shapes the library does not take)."""
    fs = [_need_cl(f, "features") for f in features]
    os_ = [_need_cl(o, "output") for o in outputs]
    bs = [_C.need_hip(b, "best_bboxes") for b in best_rbboxes]
    N, C = fs[0].shape[:2]
    for f, b, o in zip(fs, bs, os_):
        if f.shape[:2] != (N, C) or b.numel() != N * f.size(2) * f.size(3) * 5 or o.shape != f.shape:
            raise RuntimeError("levels must share N and C, bring N*H*W x 5 boxes and outputs of the features' shape")
    P = _plan(N, 0, [tuple(f.shape[2:]) for f in fs], spatial_scales, points)
    H, W, sc = P.H, P.W, P.sc
    with torch.cuda.device(fs[0].device):
        if tables is not None:
            rc = _C.lib().r3det_feature_refine_forward_levels_nhwc_tab(
                len(fs), _ptr_array(fs), _ptr_array(bs), N, C, H, W, sc, int(points), _ptr_array(os_),
                _ptr_array(tables), _C.stream())
        else:
            rc = _C.lib().r3det_feature_refine_forward_levels_nhwc(len(fs), _ptr_array(fs), _ptr_array(bs), N, C, H, W,
                                                                   sc, int(points), _ptr_array(os_), _C.stream())
    return _taken(rc, "fr_forward_levels_nhwc")


def fr_module_levels_nhwc(conv_a, conv_b, bias_a, bias_b, residual, best_rbboxes, spatial_scales, points, outputs,
                          tables=None):
    """The FeatureRefineModule tail of all pyramid levels in one library call
    (r3det_feature_refine_module_levels_nhwc); per level as ``fr_module_nhwc``.  ``conv_b`` may be None.  ``tables``
    (points = 1): per level a tap table (``tap_tables``) the launches also write."""
    as_ = [_need_cl(t, "conv_a") for t in conv_a]
    rs = [_need_cl(t, "residual") for t in residual]
    os_ = [_need_cl(t, "output") for t in outputs]
    cbs = [_need_cl(t, "conv_b") for t in conv_b] if conv_b is not None else None
    bs = [_C.need_hip(b, "best_bboxes") for b in best_rbboxes]
    N, C = as_[0].shape[:2]
    for i, (a, r, o, b) in enumerate(zip(as_, rs, os_, bs)):
        if a.shape[:2] != (N, C) or r.shape != a.shape or o.shape != a.shape or b.numel() != N * a.size(2) * a.size(3) * 5 \
                or (cbs is not None and cbs[i].shape != a.shape):
            raise RuntimeError("per level: conv_a, conv_b, residual and output of one shape; best_bboxes N*H*W x 5")
    for t, nm in ((bias_a, "bias_a"), (bias_b, "bias_b")):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or t.numel() != C or not t.is_contiguous()):
            raise RuntimeError(f"{nm} must be a contiguous fp32 CUDA tensor of C values")
    P = _plan(N, 0, [tuple(a.shape[2:]) for a in as_], spatial_scales, points)
    H, W, sc = P.H, P.W, P.sc
    with torch.cuda.device(as_[0].device):
        if tables is not None:
            rc = _C.lib().r3det_feature_refine_module_levels_nhwc_tab(
                len(as_), _ptr_array(as_), _ptr_array(cbs) if cbs is not None else None, _C.ptr(bias_a), _C.ptr(bias_b),
                _ptr_array(rs), _ptr_array(bs), N, C, H, W, sc, int(points), _ptr_array(os_), _ptr_array(tables),
                _C.stream())
        else:
            rc = _C.lib().r3det_feature_refine_module_levels_nhwc(
                len(as_), _ptr_array(as_), _ptr_array(cbs) if cbs is not None else None, _C.ptr(bias_a), _C.ptr(bias_b),
                _ptr_array(rs), _ptr_array(bs), N, C, H, W, sc, int(points), _ptr_array(os_), _C.stream())
    return _taken(rc, "fr_module_levels_nhwc")


class _LevelsPlan:
    """What a levels call needs besides the tensors, computed once per (N, C, shapes, scales, points): the ctypes shape
    arrays and the library's workspace sizes (every call of the training nodes rebuilt its ctypes arrays
    and asked the library for three sizes)."""
    __slots__ = ("n", "H", "W", "sc", "shapes", "module_ws", "bwd_ws", "bwd_nhwc_ws")

    def __init__(self, N, C, shapes, scales, points):
        import ctypes
        n = self.n = len(shapes)
        arr_i = ctypes.c_int * n
        self.shapes = tuple(shapes)
        self.H, self.W = arr_i(*[h for h, _ in shapes]), arr_i(*[w for _, w in shapes])
        self.sc = (ctypes.c_float * n)(*[float(s) for s in scales])
        L = _C.lib()
        self.module_ws = int(L.r3det_fr_module_levels_workspace_bytes(n, N, self.H, self.W))
        self.bwd_ws = int(L.r3det_fr_backward_levels_workspace_bytes(n, N, self.H, self.W, int(points)))
        self.bwd_nhwc_ws = int(L.r3det_fr_backward_nhwc_levels_workspace_bytes(n, N, self.H, self.W, int(points)))


_PLANS = {}


def _plan(N, C, shapes, scales, points):
    key = (N, C, tuple(shapes), tuple(float(s) for s in scales), int(points))
    p = _PLANS.get(key)
    if p is None:
        if len(_PLANS) >= 32:          # (a handful of pyramids per process: drop everything rather than track use)
            _PLANS.clear()
        p = _PLANS[key] = _LevelsPlan(N, C, shapes, scales, points)
    return p


def fr_module_levels(conv_a, conv_b, residual, best_rbboxes, spatial_scales, points, outputs):
    """The FeatureRefineModule tail of all NCHW pyramid levels in one library call
    (r3det_feature_refine_module_levels): ``outputs[l] = residual[l] + fr(conv_a[l] + conv_b[l], boxes[l])``, the two
    elementwise passes of feature_refine_module.py:121-126 folded into the sampler launches -- 3 launches for a 1024^2
    pyramid.  -> the workspace (it holds the tap tables of the 128 / 64 levels), or None when the library does not take
    these levels in this form (nothing was launched)."""
    if int(points) != 1:
        return None
    as_ = [_C.need_hip(t, "conv_a") for t in conv_a]
    bs_ = [_C.need_hip(t, "conv_b") for t in conv_b]
    rs = [_C.need_hip(t, "residual") for t in residual]
    os_ = [_C.need_hip(t, "output") for t in outputs]
    bx = [_C.need_hip(t, "best_bboxes") for t in best_rbboxes]
    N, C = as_[0].shape[:2]
    for a, b, r, o, q in zip(as_, bs_, rs, os_, bx):
        if a.shape[:2] != (N, C) or b.shape != a.shape or r.shape != a.shape or o.shape != a.shape or \
                q.numel() != N * a.size(2) * a.size(3) * 5:
            raise RuntimeError("per level: conv_a, conv_b, residual and output of one shape; best_bboxes N*H*W x 5")
    P = _plan(N, 0, [tuple(a.shape[2:]) for a in as_], spatial_scales, 1)
    with torch.cuda.device(as_[0].device):
        ws = torch.empty(max(P.module_ws, 16), dtype=torch.uint8, device=as_[0].device)
        rc = _C.lib().r3det_feature_refine_module_levels(
            P.n, _ptr_array(as_), _ptr_array(bs_), _ptr_array(rs), _ptr_array(bx), N, C, P.H, P.W, P.sc, 1,
            _ptr_array(os_), _C.ptr(ws), P.module_ws, _C.stream())
    return ws if _taken(rc, "fr_module_levels") else None


def fr_backward(top_grad, best_rbboxes, spatial_scale, points, bottom_grad, overwrite=False):
    """feature_refine_cuda.backward (feature_refine_cuda.cpp:44-66): accumulates into
    ``bottom_grad`` (``overwrite=True``: writes it, no zero-fill needed)."""
    g = _C.need_hip(top_grad, "top_grad")
    b = _C.need_hip(best_rbboxes, "best_bboxes")
    o = _C.need_hip(bottom_grad, "bottom_grad")
    N, C, H, W = g.shape
    with torch.cuda.device(g.device):
        L = _C.lib()
        wsb = int(L.r3det_fr_backward_workspace_bytes(N, H, W, int(points)))
        ws = torch.empty(wsb, dtype=torch.uint8, device=g.device)
        _C.check(L.r3det_feature_refine_backward_ws(_C.ptr(g), _C.ptr(b), N, C, H, W, float(spatial_scale),
                                                    int(points), _C.ptr(o), int(bool(overwrite)), _C.ptr(ws), wsb,
                                                    _C.stream()), "fr_backward")
    return 1


def fr_backward_levels(top_grads, best_rbboxes, spatial_scales, points, bottom_grads, overwrite=True):
    """The backward of all pyramid levels (NCHW) in two library calls: the indexes of the levels' boxes from one grouped
    launch (r3det_feature_refine_backward_index_levels), then the gathers
    (r3det_feature_refine_backward_levels_indexed).  ``overwrite=False`` accumulates into ``bottom_grads`` as the
    reference's backward does (feature_refine_cuda.cpp:44-66)."""
    gs = [_C.need_hip(g, "top_grad") for g in top_grads]
    os_ = [_C.need_hip(o, "bottom_grad") for o in bottom_grads]
    bs = [_C.need_hip(b, "best_bboxes") for b in best_rbboxes]
    N, C = gs[0].shape[:2]
    H, W, sc, _ = _lvl_arrays(gs, spatial_scales)
    L = _C.lib()
    n = len(gs)
    with torch.cuda.device(gs[0].device):
        wsb = int(L.r3det_fr_backward_levels_workspace_bytes(n, N, H, W, int(points)))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=gs[0].device)
        _C.check(L.r3det_feature_refine_backward_index_levels(n, _ptr_array(bs), N, C, H, W, sc, int(points), _C.ptr(ws),
                                                              wsb, _C.stream()), "fr_backward_index_levels")
        _C.check(L.r3det_feature_refine_backward_levels_indexed(n, _ptr_array(gs), _ptr_array(bs), N, C, H, W, sc,
                                                                int(points), _ptr_array(os_), int(bool(overwrite)),
                                                                _C.ptr(ws), wsb, _C.stream()),
                 "fr_backward_levels_indexed")
    return 1


def fr_backward_levels_nhwc(top_grads, best_rbboxes, spatial_scales, points, bottom_grads, overwrite=True):
    """The channels_last backward of all pyramid levels in two library calls
    (r3det_feature_refine_backward_nhwc_index_levels: the CSR indexes of all levels from one grouped launch, then
    r3det_feature_refine_backward_nhwc_levels_indexed: the gathers, the coarse levels one grid).  False when a level
    has no workspace form: nothing was launched."""
    gs = [_need_cl(g, "top_grad") for g in top_grads]
    os_ = [_need_cl(o, "bottom_grad") for o in bottom_grads]
    bs = [_C.need_hip(b, "best_bboxes") for b in best_rbboxes]
    N, C = gs[0].shape[:2]
    H, W, sc, _ = _lvl_arrays(gs, spatial_scales)
    L = _C.lib()
    n = len(gs)
    with torch.cuda.device(gs[0].device):
        wsb = int(L.r3det_fr_backward_nhwc_levels_workspace_bytes(n, N, H, W, int(points)))
        if wsb == 0:
            return False
        ws = torch.empty(wsb, dtype=torch.uint8, device=gs[0].device)
        _C.check(L.r3det_feature_refine_backward_nhwc_index_levels(n, _ptr_array(bs), N, H, W, sc, int(points), _C.ptr(ws),
                                                                   wsb, _C.stream()), "fr_backward_nhwc_index_levels")
        _C.check(L.r3det_feature_refine_backward_nhwc_levels_indexed(n, _ptr_array(gs), N, C, H, W, int(points),
                                                                     _ptr_array(os_), int(bool(overwrite)), _C.ptr(ws),
                                                                     wsb, _C.stream()),
                 "fr_backward_nhwc_levels_indexed")
    return True


def fr_backward_nhwc(top_grad, best_rbboxes, spatial_scale, points, bottom_grad, overwrite=False, index=None):
    """feature_refine_cuda.backward on channels_last memory (r3det_feature_refine_backward_nhwc): ``top_grad`` /
    ``bottom_grad`` are (N, C, H, W) tensors in torch.channels_last.  A gather over the inverse tap index of the
    boxes: no atomics, one summation order.  ``index``: result of ``fr_backward_nhwc_index`` for these boxes (then
    the boxes are not read).  False when the library does not take the shape: nothing was launched."""
    g, o = _need_cl(top_grad, "top_grad"), _need_cl(bottom_grad, "bottom_grad")
    N, C, H, W = g.shape
    if o.shape != g.shape:
        raise RuntimeError("bottom_grad must have top_grad's shape")
    L = _C.lib()
    with torch.cuda.device(g.device):
        if index is not None:
            rc = L.r3det_feature_refine_backward_nhwc_indexed(_C.ptr(g), N, C, H, W, int(points), _C.ptr(o),
                                                              int(bool(overwrite)), _C.ptr(index), index.numel(),
                                                              _C.stream())
            return _taken(rc, "fr_backward_nhwc_indexed")
        b = _C.need_hip(best_rbboxes, "best_bboxes")
        if b.numel() != N * H * W * 5:
            raise RuntimeError(f"best_bboxes must hold N*H*W x 5 values, got {tuple(b.shape)}")
        wsb = int(L.r3det_fr_backward_nhwc_workspace_bytes(N, H, W, int(points)))
        if wsb == 0:
            return False
        ws = torch.empty(wsb, dtype=torch.uint8, device=g.device)
        rc = L.r3det_feature_refine_backward_nhwc(_C.ptr(g), _C.ptr(b), N, C, H, W, float(spatial_scale), int(points),
                                                  _C.ptr(o), int(bool(overwrite)), _C.ptr(ws), wsb, _C.stream())
    return _taken(rc, "fr_backward_nhwc")


def fr_backward_nhwc_index(best_rbboxes, N, H, W, spatial_scale, points=1):
    """The inverse tap index of one level's boxes (r3det_feature_refine_backward_nhwc_index): depends on the boxes
    only, so it can be built when the forward pass has them.  None when the shape is not taken."""
    b = _C.need_hip(best_rbboxes, "best_bboxes")
    L = _C.lib()
    wsb = int(L.r3det_fr_backward_nhwc_workspace_bytes(N, H, W, int(points)))
    if wsb == 0 or b.numel() != N * H * W * 5:
        return None
    with torch.cuda.device(b.device):
        ws = torch.empty(wsb, dtype=torch.uint8, device=b.device)
        rc = L.r3det_feature_refine_backward_nhwc_index(_C.ptr(b), N, H, W, float(spatial_scale), int(points),
                                                        _C.ptr(ws), wsb, _C.stream())
    return ws if _taken(rc, "fr_backward_nhwc_index") else None


def fr_backward_nhwc_index_levels(best_rbboxes, N, shapes, spatial_scales, points=1, tables=None):
    """The CSR indexes of all levels for the channels_last gathers (r3det_feature_refine_backward_nhwc_index_levels;
    with ``tables`` -- per level a tap table of the same boxes or None -- the _tab form, whose scan reads the tables).
    -> (workspace, bytes) for ``r3det_feature_refine_backward_nhwc_levels_indexed``, or None (a level has no form)."""
    bs = [_C.need_hip(b, "best_bboxes") for b in best_rbboxes]
    P = _plan(N, 0, shapes, spatial_scales, points)
    wsb = P.bwd_nhwc_ws
    if wsb == 0:
        return None
    L = _C.lib()
    with torch.cuda.device(bs[0].device):
        ws = torch.empty(wsb, dtype=torch.uint8, device=bs[0].device)
        if tables is not None and int(points) == 1 and any(t is not None for t in tables):
            _C.check(L.r3det_feature_refine_backward_nhwc_index_levels_tab(
                P.n, _ptr_array(bs), _ptr_array(tables), N, P.H, P.W, P.sc, 1, _C.ptr(ws), wsb, _C.stream()),
                "fr_backward_nhwc_index_levels_tab")
        else:
            _C.check(L.r3det_feature_refine_backward_nhwc_index_levels(
                P.n, _ptr_array(bs), N, P.H, P.W, P.sc, int(points), _C.ptr(ws), wsb, _C.stream()),
                "fr_backward_nhwc_index_levels")
    return ws, wsb


def fr_backward_index_levels(best_rbboxes, N, C, shapes, spatial_scales, points=1, tables=None):
    """The indexes of all levels for the NCHW gathers (r3det_feature_refine_backward_index_levels / _tab as above;
    ``fr_prepare``'s table of a level is a tap table).  -> (workspace, bytes)."""
    bs = [_C.need_hip(b, "best_bboxes") for b in best_rbboxes]
    P = _plan(N, 0, shapes, spatial_scales, points)
    wsb = P.bwd_ws
    L = _C.lib()
    with torch.cuda.device(bs[0].device):
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=bs[0].device)
        if tables is not None and int(points) == 1 and any(t is not None for t in tables):
            _C.check(L.r3det_feature_refine_backward_index_levels_tab(
                P.n, _ptr_array(bs), _ptr_array(tables), N, C, P.H, P.W, P.sc, 1, _C.ptr(ws), wsb, _C.stream()),
                "fr_backward_index_levels_tab")
        else:
            _C.check(L.r3det_feature_refine_backward_index_levels(
                P.n, _ptr_array(bs), N, C, P.H, P.W, P.sc, int(points), _C.ptr(ws), wsb, _C.stream()),
                "fr_backward_index_levels")
    return ws, wsb


# training: the forward launches also write the levels' tap tables and the backward's index is built from them (round
# 6; False: the index kernels scan the box records, rounds 3-5 -- tests and tools flip it for the A/B)
TRAIN_TAP_TABLES = True
NHWC_ONLY = False  # tests: fail instead of falling back when a channels_last module input does not take the NHWC launch
# training: FeatureRefineModule's tail (add, samplers, residual add) as one autograd node (False: the three-step form of
# rounds 1-4 around FeatureRefineLevelsFunction; tests and tools/train_hot_path.py flip it for the A/B)
TRAIN_FUSED_TAIL = True


def fr_backward_index(best_rbboxes, N, C, H, W, spatial_scale, points=1):
    """The inverse tap index of one level's boxes for the NCHW backward of an (N, C, H, W) gradient
    (r3det_feature_refine_backward_index): depends on the boxes only, so the forward pass builds it
    (feature_refine_module.py:18-26 saves the boxes there) and the backward proper is the gather alone.  None when
    the shape has no gather form."""
    b = _C.need_hip(best_rbboxes, "best_bboxes")
    L = _C.lib()
    wsb = int(L.r3det_fr_backward_workspace_bytes(N, H, W, int(points)))
    if wsb == 0 or b.numel() != N * H * W * 5:
        return None
    with torch.cuda.device(b.device):
        ws = torch.empty(wsb, dtype=torch.uint8, device=b.device)
        rc = L.r3det_feature_refine_backward_index(_C.ptr(b), N, C, H, W, float(spatial_scale), int(points),
                                                   _C.ptr(ws), wsb, _C.stream())
    return ws if _taken(rc, "fr_backward_index") else None


def fr_backward_indexed(top_grad, points, bottom_grad, index, overwrite=True):
    """The gather alone over an index made by ``fr_backward_index`` (r3det_feature_refine_backward_indexed);
    False when the library has no gather form for this (shape, C): nothing was launched."""
    g = _C.need_hip(top_grad, "top_grad")
    o = _C.need_hip(bottom_grad, "bottom_grad")
    N, C, H, W = g.shape
    if o.shape != g.shape:
        raise RuntimeError("bottom_grad must have top_grad's shape")
    with torch.cuda.device(g.device):
        rc = _C.lib().r3det_feature_refine_backward_indexed(_C.ptr(g), N, C, H, W, int(points), _C.ptr(o),
                                                            int(bool(overwrite)), _C.ptr(index), index.numel(),
                                                            _C.stream())
    return _taken(rc, "fr_backward_indexed")


class FeatureRefineFunction(Function):
    """autograd wrapper (feature_refine_module.py:10-40); no gradient flows to the boxes.  The backward's index of
    the boxes is built here in ``forward``, where the boxes are at hand, so ``backward`` is one gather launch."""

    @staticmethod
    def forward(ctx, features, best_rbboxes, spatial_scale, points=1, table=None):
        ctx.spatial_scale = spatial_scale
        ctx.points = points
        ctx.save_for_backward(best_rbboxes)
        assert points in [1, 5]
        assert features.is_cuda
        ctx.index = None
        ctx.nhwc = False
        N, C, H, W = features.shape
        boxes = best_rbboxes.contiguous()
        if _is_cl(features):
            # channels_last pipelines (training included): sampler and its backward on (N, H, W, C) memory, no
            # layout switch around them
            output = torch.empty_like(features)  # (preserves channels_last)
            if ctx.needs_input_grad[0] and points == 1 and TRAIN_TAP_TABLES:
                # the sampler launch also writes the level's tap table; the backward's index is built from it
                tabs = tap_tables(N, [(H, W)], features.device)
                if fr_forward_levels_nhwc([features], [boxes], [spatial_scale], 1, [output], tabs):
                    ctx.nhwc = True
                    idx = fr_backward_nhwc_index_levels([boxes], N, [(H, W)], [spatial_scale], 1, tabs)
                    ctx.index = idx[0] if idx is not None else None
                    return output
            if fr_forward_nhwc(features, boxes, spatial_scale, points, output):
                ctx.nhwc = True
                if ctx.needs_input_grad[0]:
                    ctx.index = fr_backward_nhwc_index(boxes, N, H, W, spatial_scale, points)
                return output
        features = features.contiguous()
        output = torch.empty_like(features)  # the kernel overwrites every element
        prepared = table is not None and fr_forward_prepared(features, table, output)
        if not prepared:
            fr_forward(features, boxes, spatial_scale, points, output)
        if ctx.needs_input_grad[0]:
            if prepared and points == 1 and TRAIN_TAP_TABLES and \
                    int(_C.lib().r3det_fr_backward_workspace_bytes(N, H, W, 1)):
                # (the caller's table is the level's tap table: the index kernel scans it instead of the box records)
                ctx.index = fr_backward_index_levels([boxes], N, C, [(H, W)], [spatial_scale], 1, [table])[0]
            else:
                ctx.index = fr_backward_index(boxes, N, C, H, W, spatial_scale, points)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        best_rbboxes = ctx.saved_tensors[0]
        assert grad_output.is_cuda
        grad_input = None
        if ctx.needs_input_grad[0]:
            if ctx.nhwc:
                g = grad_output.contiguous(memory_format=torch.channels_last)
                grad_input = torch.empty_like(g)
                if fr_backward_nhwc(g, best_rbboxes.contiguous(), ctx.spatial_scale, ctx.points, grad_input,
                                    overwrite=True, index=ctx.index):
                    return grad_input, None, None, None, None
            grad_output = grad_output.contiguous()
            grad_input = torch.empty_like(grad_output)
            if ctx.nhwc or ctx.index is None or not fr_backward_indexed(grad_output, ctx.points, grad_input, ctx.index):
                fr_backward(grad_output, best_rbboxes.contiguous(), ctx.spatial_scale, ctx.points, grad_input,
                            overwrite=True)
        return grad_input, None, None, None, None


feature_refine = FeatureRefineFunction.apply


def _levels_gather(ctx, grads):
    """The gathers of all levels over the index the forward pass built (both levels nodes' backward): -> (the incoming
    gradients in the node's layout, the gathered gradients)."""
    n, boxes = ctx.n, ctx.saved_tensors
    fmt = torch.channels_last if ctx.nhwc else torch.contiguous_format
    gs = []
    for g, shp in zip(grads, ctx.shape):
        if g is None:
            g = torch.zeros(shp, device=boxes[0].device)
        gs.append(g if g.is_contiguous(memory_format=fmt) else g.contiguous(memory_format=fmt))
    ds = [torch.empty_like(g) for g in gs]  # (preserves the layout)
    N, C = ctx.shape[0][:2]
    P = _plan(N, 0, [shp[2:] for shp in ctx.shape], ctx.scales, ctx.points)
    L = _C.lib()
    ws, wsb = ctx.index
    with torch.cuda.device(gs[0].device):
        if ctx.nhwc:
            _C.check(L.r3det_feature_refine_backward_nhwc_levels_indexed(
                n, _ptr_array(gs), N, C, P.H, P.W, int(ctx.points), _ptr_array(ds), 1, _C.ptr(ws), wsb, _C.stream()),
                "fr_backward_nhwc_levels_indexed")
        else:
            _C.check(L.r3det_feature_refine_backward_levels_indexed(
                n, _ptr_array(gs), _ptr_array(boxes), N, C, P.H, P.W, P.sc, int(ctx.points), _ptr_array(ds), 1,
                _C.ptr(ws), wsb, _C.stream()), "fr_backward_levels_indexed")
    return gs, ds


class FeatureRefineLevelsFunction(Function):
    """The sampler of ALL pyramid levels as one autograd node: one library call for the samplers
    (r3det_feature_refine_forward_levels), one for the backward's indexes of the boxes -- built here, where the boxes
    are at hand -- and one for the gathers (r3det_feature_refine_backward_index_levels / _levels_indexed); one
    workspace each.  Channels_last inputs (all levels) stay on (N, H, W, C) memory: the _nhwc forms of the three calls.
    The reference runs one FeatureRefineFunction per level (feature_refine_module.py:108-127); from Python every level
    costs ~10 us of host time per call, more than the kernels of the three coarse levels."""

    @staticmethod
    def forward(ctx, spatial_scales, points, n, *tensors):
        boxes = [t.contiguous() for t in tensors[n:2 * n]]
        assert points in [1, 5] and all(f.is_cuda for f in tensors[:n])
        N, C = tensors[0].shape[:2]
        ctx.scales, ctx.points, ctx.n = list(spatial_scales), points, n
        ctx.shape = [tuple(f.shape) for f in tensors[:n]]
        ctx.save_for_backward(*boxes)
        ctx.index = None
        ctx.nhwc = False
        if all(_is_cl(f) for f in tensors[:n]) and C % 4 == 0:
            feats = list(tensors[:n])
            shapes = [tuple(f.shape[2:]) for f in feats]
            wsb = _plan(N, 0, shapes, spatial_scales, points).bwd_nhwc_ws
            outs = [torch.empty_like(f) for f in feats]  # (preserves channels_last)
            need = any(ctx.needs_input_grad[3:3 + n])
            # the sampler launches also write the levels' tap tables when a gradient will be asked for
            tables = tap_tables(N, shapes, feats[0].device) if (need and wsb and points == 1 and TRAIN_TAP_TABLES) else None
            if wsb and fr_forward_levels_nhwc(feats, boxes, spatial_scales, points, outs, tables):
                ctx.nhwc = True
                if need:
                    ctx.index = fr_backward_nhwc_index_levels(boxes, N, shapes, spatial_scales, points, tables)
                return tuple(outs)
        feats = [t.contiguous() for t in tensors[:n]]
        outs = [torch.empty_like(f) for f in feats]
        fr_forward_levels(feats, boxes, spatial_scales, points, outs)
        if any(ctx.needs_input_grad[3:3 + n]):
            ctx.index = fr_backward_index_levels(boxes, N, C, [tuple(f.shape[2:]) for f in feats], spatial_scales, points)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        n, boxes = ctx.n, ctx.saved_tensors
        if ctx.index is None:
            return (None, None, None) + (None,) * (2 * n)
        _, outs = _levels_gather(ctx, grads)
        return (None, None, None) + tuple(outs) + (None,) * n


class FeatureRefineModuleLevelsFunction(Function):
    """The TAIL of FeatureRefineModule for all pyramid levels as one autograd node (round 5, SURVEY 8f rank 2 for the
    training step): ``out_l = x_l + fr(a_l + b_l, boxes_l)`` with a_l = conv_5_1(conv_1_5(x_l)), b_l = conv_1_1(x_l)
    (feature_refine_module.py:120-126).  Forward = the launches inference uses -- the add in front of the sampler and
    the residual add behind it ride in the sampler launch (r3det_feature_refine_module_levels_nhwc on channels_last
    memory; r3det_feature_refine_module_prepared per NCHW level where it takes the shape, the three-step form
    elsewhere) -- instead of two elementwise passes per level around FeatureRefineLevelsFunction.  Backward = the
    gather over the inverse tap index built in the forward pass: d = fr_backward(g) is the gradient of BOTH a and b
    (their sum is the sampler's input), g itself the gradient of x; no elementwise pass either side."""

    @staticmethod
    def forward(ctx, spatial_scales, points, n, *tensors):
        a_in, b_in, x_in = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        boxes = [t.contiguous() for t in tensors[3 * n:4 * n]]
        assert points in [1, 5] and all(f.is_cuda for f in x_in)
        N, C = x_in[0].shape[:2]
        ctx.scales, ctx.points, ctx.n = list(spatial_scales), points, n
        ctx.shape = [tuple(f.shape) for f in x_in]
        ctx.save_for_backward(*boxes)
        ctx.index, ctx.nhwc = None, False
        need = any(ctx.needs_input_grad[3:3 + 3 * n])
        if all(_is_cl(f) for f in a_in + b_in + x_in) and C % 4 == 0:
            shapes = [tuple(f.shape[2:]) for f in x_in]
            wsb = _plan(N, 0, shapes, spatial_scales, points).bwd_nhwc_ws
            outs = [torch.empty_like(f) for f in x_in]  # (preserves channels_last)
            # the module launches also write the levels' tap tables when a gradient will be asked for: the index
            # kernel below scans those (4 contiguous bytes per source) instead of the box records
            tables = tap_tables(N, shapes, x_in[0].device) if (need and wsb and points == 1 and TRAIN_TAP_TABLES) else None
            if wsb and fr_module_levels_nhwc(list(a_in), list(b_in), None, None, list(x_in), boxes, spatial_scales,
                                             points, outs, tables):
                ctx.nhwc = True
                if need:
                    ctx.index = fr_backward_nhwc_index_levels(boxes, N, shapes, spatial_scales, points, tables)
                return tuple(outs)
        a_c, b_c, x_c = ([t.contiguous() for t in ts] for ts in (a_in, b_in, x_in))
        outs = [torch.empty_like(x) for x in x_c]
        # one library call for the whole tail (round 6): the coarse levels one grid that also builds the cell levels'
        # tap tables, one fused launch per cell level -- where rounds 3-5 looped over the levels from Python
        if fr_module_levels(a_c, b_c, x_c, boxes, spatial_scales, points, outs) is None:
            for a, b, x, bx, s, o in zip(a_c, b_c, x_c, boxes, spatial_scales, outs):
                table = fr_prepare(bx, x.size(0), x.size(2), x.size(3), s, points)
                if table is None or not fr_module_prepared(a, b, x, table, o):
                    m = a + b
                    fr_forward(m, bx, s, points, o)
                    o += x
        if need:
            # (the NCHW gathers' index scans the box records: its SELL form gains nothing from the tap tables --
            # 13.4 against 13.5 us at level 0, profiles/r06_fr_backward_index_ab.txt)
            ctx.index = fr_backward_index_levels(boxes, N, C, [tuple(f.shape[2:]) for f in x_c], spatial_scales, points)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        n, boxes = ctx.n, ctx.saved_tensors
        if ctx.index is None:
            return (None, None, None) + (None,) * (4 * n)
        gs, ds = _levels_gather(ctx, grads)
        ds = tuple(ds)
        return (None, None, None) + ds + ds + tuple(gs) + (None,) * n


def feature_refine_module_levels(conv_a, conv_b, x, best_rbboxes, spatial_scales, points=1):
    """``[x_l + feature_refine(a_l + b_l, boxes_l, s_l, points)]`` over the pyramid as ONE autograd node."""
    n = len(x)
    return list(FeatureRefineModuleLevelsFunction.apply(tuple(float(s) for s in spatial_scales), points, n, *conv_a,
                                                        *conv_b, *x, *best_rbboxes))


def feature_refine_levels(features, best_rbboxes, spatial_scales, points=1):
    """``[feature_refine(f, b, s, points) for f, b, s in zip(...)]`` as one autograd node (levels sharing N and C;
    NCHW, or channels_last when every level is)."""
    n = len(features)
    return list(FeatureRefineLevelsFunction.apply(tuple(float(s) for s in spatial_scales), points, n, *features,
                                                  *best_rbboxes))




class FR(nn.Module):
    """One pyramid level's sampler (feature_refine_module.py:46-63)."""

    def __init__(self, spatial_scale, points=1):
        super().__init__()
        self.spatial_scale = float(spatial_scale)
        self.points = points

    def forward(self, features, best_rbboxes, table=None):
        """``table``: optional result of ``fr_prepare`` for these boxes (not in the reference's signature)."""
        return feature_refine(features, best_rbboxes, self.spatial_scale, self.points, table)

    def __repr__(self):
        return f'{self.__class__.__name__}(spatial_scale={self.spatial_scale}, points={self.points})'


class FeatureRefineModule(nn.Module):
    """conv(5x1 o 1x5) + conv1x1 -> FR sampler -> residual, per level
    (feature_refine_module.py:66-127)."""

    def __init__(self, in_channels, featmap_strides, conv_cfg=None, norm_cfg=None):
        super().__init__()
        self.in_channels = in_channels
        self.featmap_strides = featmap_strides
        self.conv_cfg = conv_cfg
        self.norm_cfg = norm_cfg
        self._init_layers()

    def _init_layers(self):
        c = self.in_channels
        self.fr = nn.ModuleList([FR(spatial_scale=1 / s) for s in self.featmap_strides])
        self.conv_5_1 = nn.Conv2d(c, c, kernel_size=(5, 1), stride=1, padding=(2, 0))
        self.conv_1_5 = nn.Conv2d(c, c, kernel_size=(1, 5), stride=1, padding=(0, 2))
        self.conv_1_1 = nn.Conv2d(c, c, kernel_size=1)

    def init_weights(self):
        # mmcv.cnn.normal_init(m, std=0.01): weight ~ N(0, 0.01), bias = 0
        for m in (self.conv_5_1, self.conv_1_5, self.conv_1_1):
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0)

    def forward(self, x, best_rbboxes):
        """x: list of per-level (N,C,H,W); best_rbboxes: list over images of lists over levels
        of (H*W, 5)."""
        per_level = [torch.cat(lvl) for lvl in zip(*best_rbboxes)]
        # the fused forward-only launches build no autograd graph: only when nothing here can need a gradient
        no_grad = not (torch.is_grad_enabled() and (any(f.requires_grad for f in x)
                                                    or any(p.requires_grad for p in self.parameters())))

        is_cl = _is_cl
        nhwc = [no_grad and is_cl(f) for f in x]
        if not no_grad and len({f.shape[:2] for f in x}) == 1 and len({fr.points for fr in self.fr}) == 1:
            # training: the samplers of all levels as ONE autograd node (one library call each for the samplers, the
            # backward's indexes and the gathers) -- on channels_last memory when every level's convolution output is,
            # else on NCHW planes
            if TRAIN_FUSED_TAIL:
                # round 5: the add in front of the samplers and the residual add behind them inside the node (the
                # forward is the inference launch, the backward the gather alone: FeatureRefineModuleLevelsFunction)
                return feature_refine_module_levels([self.conv_5_1(self.conv_1_5(f)) for f in x],
                                                    [self.conv_1_1(f) for f in x], list(x),
                                                    [b.contiguous() for b in per_level],
                                                    [fr.spatial_scale for fr in self.fr], self.fr[0].points)
            mixed = [self.conv_5_1(self.conv_1_5(f)) + self.conv_1_1(f) for f in x]
            sampled = feature_refine_levels(mixed, [b.contiguous() for b in per_level],
                                            [fr.spatial_scale for fr in self.fr], self.fr[0].points)
            return [f + o for f, o in zip(x, sampled)]
        if all(nhwc) and len({f.shape[:2] for f in x}) == 1 and len({fr.points for fr in self.fr}) == 1:
            # channels_last inference: the module tail of ALL levels in one library call -- level 0 (the wide regions
            # form) one launch, the coarse levels together one more -- on channels_last memory: the two convolutions'
            # bias adds, their sum, the sampler and the residual add (3 reads + 1 write per element); the results feed
            # the refine head's channels_last convolutions directly
            ras = [F.conv2d(self.conv_1_5(f), self.conv_5_1.weight, None, self.conv_5_1.stride, self.conv_5_1.padding)
                   for f in x]
            rbs = [F.conv2d(f, self.conv_1_1.weight, None, self.conv_1_1.stride, self.conv_1_1.padding) for f in x]
            if all(is_cl(t) for t in ras) and all(is_cl(t) for t in rbs):
                fused = [torch.empty_like(f) for f in x]  # (preserves channels_last)
                if fr_module_levels_nhwc(ras, rbs, self.conv_5_1.bias, self.conv_1_1.bias, x,
                                         [b.contiguous() for b in per_level], [fr.spatial_scale for fr in self.fr],
                                         self.fr[0].points, fused):
                    return fused
            if NHWC_ONLY:
                raise RuntimeError("channels_last FR module path not taken")
        if no_grad and not any(is_cl(f) for f in x) and len({f.shape[:2] for f in x}) == 1 and \
                all(fr.points == 1 for fr in self.fr) and all(f.is_cuda and f.dtype == torch.float32 for f in x):
            # NCHW inference: the module tail of ALL levels in one library call (r3det_feature_refine_module_levels:
            # the coarse levels one grid that also builds the 128 / 64 levels' tap tables, one fused launch each for those)
            a_all = [self.conv_5_1(self.conv_1_5(f)).contiguous() for f in x]
            b_all = [self.conv_1_1(f).contiguous() for f in x]
            x_all = [f.contiguous() for f in x]
            fused = [torch.empty_like(f) for f in x_all]
            if fr_module_levels(a_all, b_all, x_all, [b.contiguous() for b in per_level],
                                [fr.spatial_scale for fr in self.fr], 1, fused) is not None:
                return fused
        # tap tables of the NCHW levels first: each sampler call below is then a single launch with no
        # dependent launch in front of it (the channels_last launch derives its taps from the boxes itself)
        # (channels_last training levels take the NHWC sampler, which needs no table either)
        tables = [None if cl or (is_cl(f) and not no_grad)
                  else fr_prepare(b, f.size(0), f.size(2), f.size(3), fr.spatial_scale, fr.points)
                  for f, b, fr, cl in zip(x, per_level, self.fr, nhwc)]
        out = []
        for feat, boxes, fr, table, cl in zip(x, per_level, self.fr, tables, nhwc):
            if cl:
                # channels_last inference: ONE launch for the module's tail, on channels_last memory -- the two
                # convolutions' bias adds, their sum, the sampler and the residual add (3 reads + 1 write per
                # element); the result feeds the refine head's channels_last convolutions directly.  (Round 1
                # switched layouts around an NCHW sampler: two transposing passes per level.)
                ra = F.conv2d(self.conv_1_5(feat), self.conv_5_1.weight, None, self.conv_5_1.stride,
                              self.conv_5_1.padding)
                rb = F.conv2d(feat, self.conv_1_1.weight, None, self.conv_1_1.stride, self.conv_1_1.padding)
                if is_cl(ra) and is_cl(rb):
                    fused = torch.empty_like(feat)  # (preserves channels_last)
                    if fr_module_nhwc(ra, rb, self.conv_5_1.bias, self.conv_1_1.bias, feat, boxes.contiguous(),
                                      fr.spatial_scale, fr.points, fused):
                        out.append(fused)
                        continue
                if NHWC_ONLY:
                    raise RuntimeError("channels_last FR module path not taken")
            if is_cl(feat) and not no_grad:
                # channels_last training: convolutions, sampler and the sampler's backward all stay on
                # (N, H, W, C) memory (FeatureRefineFunction takes the NHWC kernels for channels_last input)
                mixed = self.conv_5_1(self.conv_1_5(feat)) + self.conv_1_1(feat)
                if is_cl(mixed):
                    out.append(feat + fr(mixed, boxes))
                    continue
            # NCHW: the sampler reads NCHW planes (no-ops for NCHW callers, like the reference)
            infer = table is not None and no_grad
            a, b = self.conv_5_1(self.conv_1_5(feat)).contiguous(), self.conv_1_1(feat).contiguous()
            feat = feat.contiguous()
            if infer:
                # inference: the add in front of the sampler and the residual add behind it ride in the
                # sampler launch (3 reads + 1 write per element instead of 8 passes over three launches)
                fused = torch.empty_like(feat)
                if fr_module_prepared(a, b, feat, table, fused):
                    out.append(fused)
                    continue
            out.append(feat + fr(a + b, boxes, table))
        return out
