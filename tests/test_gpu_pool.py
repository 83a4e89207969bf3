"""GPU: the fused per-level pre-NMS pool (r3det_level_pool: sigmoid + top-nms_pre + decode) against the op-by-op
torch form of RAnchorHead._get_bboxes_single (rotate_anchor_head.py:626-673), for NCHW and channels_last head
outputs, shared anchors (base head) and per-image rois (refine head), levels with and without a top-k."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_head(refine, strides):
    from r3det.models.heads import RRetinaHead, RRetinaRefineHead
    torch.manual_seed(1)
    h = (RRetinaRefineHead if refine else RRetinaHead)(num_classes=15, in_channels=8, feat_channels=8, strides=strides)
    return h.cuda().eval()


def spread_logits(N, A, C, H, W, seed):
    """cls maps whose best-class logit per (position, anchor) is a distinct, well separated value (a shuffled
    linspace): the top-k set and its order are then the same for any correctly rounded sigmoid."""
    g = torch.Generator().manual_seed(seed)
    L = H * W * A
    cls = torch.empty(N, L, C)
    for n in range(N):
        best = torch.linspace(-6, 4, L)[torch.randperm(L, generator=g)]
        rest = best[:, None] - 0.5 - 3 * torch.rand(L, C, generator=g)
        which = torch.randint(0, C, (L,), generator=g)
        rest[torch.arange(L), which] = best
        cls[n] = rest
    # (N, L = (p, a), C) -> (N, A*C, H, W)
    return cls.view(N, H, W, A * C).permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("refine", [False, True])
def test_pool_matches_torch_path(refine, channels_last):
    strides = (8, 16, 32)
    head = make_head(refine, strides)
    A, C, N = head.num_anchors, 15, 3
    sizes = [(64, 48), (20, 20), (5, 7)]
    cfg = dict(nms_pre=700, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
    g = torch.Generator().manual_seed(3)
    cls = [spread_logits(N, A, C, h, w, 10 + i).cuda() for i, (h, w) in enumerate(sizes)]
    reg = [(torch.randn(N, A * 5, h, w, generator=g) * torch.tensor([0.3, 0.3, 2.5, 2.5, 0.4]).repeat(A)[None, :, None, None]).cuda()
           for h, w in sizes]
    rois = None
    if refine:
        rois = [[(torch.rand(h * w, 5, generator=g) * torch.tensor([400., 300., 60., 40., 1.]) + torch.tensor([0., 0., 4., 4., -1.5])).cuda()
                 for h, w in sizes] for _ in range(N)]
    if channels_last:
        cls = [c.contiguous(memory_format=torch.channels_last) for c in cls]
        reg = [r.contiguous(memory_format=torch.channels_last) for r in reg]
    img_shape = (300, 420)
    with torch.no_grad():
        gb, gs = head.decode_bboxes(cls, reg, img_shape, cfg, rois=rois)
        wb, ws = head.decode_bboxes_torch(cls, reg, img_shape, cfg, rois=rois)
    assert gb.shape == wb.shape and gs.shape == ws.shape
    assert gs.shape[1] == 700 + (700 if 400 * A > 700 else 400 * A) + 35 * A
    assert torch.allclose(gs, ws, rtol=0, atol=1e-6), float((gs - ws).abs().max())
    assert torch.allclose(gb, wb, rtol=1e-6, atol=1e-4), float((gb - wb).abs().max())
    # inside a level with a top-k the rows come in descending order of the best class score
    top = gs[:, :700, :-1].max(2)[0]
    assert bool((top[:, 1:] <= top[:, :-1]).all())
    assert float(gs[..., -1].abs().max()) == 0


def test_pool_ties_take_lowest_rows_and_large_level():
    """Every score equal: the k winners are the k lowest rows (this build's rule; torch.topk leaves ties open).
    And a level of 147 456 rows (RRetinaNet level 0) against torch on well separated scores."""
    from r3det.ops import fr_boxes
    N, A, C, H, W, k = 2, 9, 15, 16, 16, 500
    cls = torch.zeros(N, A * C, H, W, device='cuda')
    reg = torch.zeros(N, A * 5, H, W, device='cuda')
    anchors = torch.rand(H * W * A, 5, device='cuda') * 50 + 5
    boxes = torch.full((N, k, 5), float('nan'), device='cuda')
    scores = torch.full((N, k, C + 1), float('nan'), device='cuda')
    assert fr_boxes.level_pool(cls, reg, anchors, A, C, k, None, boxes, scores, 0) == k
    assert torch.equal(boxes[0], anchors[:k]) and torch.equal(boxes[1], anchors[:k])
    assert torch.equal(scores[..., :-1], torch.full((N, k, C), 0.5, device='cuda'))
    head = make_head(False, (8,))
    A = head.num_anchors
    cls = [spread_logits(1, A, C, 128, 128, 5).cuda()]
    reg = [torch.randn(1, A * 5, 128, 128, device='cuda') * 0.2]
    cfg = dict(nms_pre=2000)
    with torch.no_grad():
        gb, gs = head.decode_bboxes(cls, reg, (1024, 1024), cfg)
        wb, ws = head.decode_bboxes_torch(cls, reg, (1024, 1024), cfg)
    assert gb.shape == (1, 2000, 5)
    assert torch.allclose(gs, ws, rtol=0, atol=1e-6) and torch.allclose(gb, wb, rtol=1e-6, atol=1e-4)


def test_pool_argument_errors():
    from r3det import _C
    L = _C.lib()
    assert L.r3det_level_pool(None, None, None, None, None, 0, 1, 1, 1, 1, 1, 1, 1.0, -1.0, -1.0, None, None, 1, 0, None, 0,
                              None) == -1
    lists = (4096 + 8192) * 8 + 16  # per image: the entries above / inside the threshold bin, 4 ints
    assert L.r3det_level_pool_workspace_bytes(2, 9, 128, 128, 2000) == 2 * 147456 * 4 + 2 * lists + 768 + 2 * 4096 * 4
    assert L.r3det_level_pool_workspace_bytes(1, 9, 5, 7, 100) == 316 * 4 + lists + 768 + 4096 * 4  # keys padded to a multiple of 4; + histogram
    assert L.r3det_level_pool_workspace_bytes(2, 1, 8, 8, 2000) == 0


def test_pool_flat_scores_large_level_and_largest_k():
    """147 456 equal scores: every key survives the first three digits, more than the LDS copy holds, so all eight
    digits run on global memory (the fallback of the select kernel); and nms_pre = 4096, the largest the library
    takes (two compare-exchange pairs per thread in the sort), on separated scores against torch.topk."""
    from r3det.ops import fr_boxes
    N, A, C, H, W, k = 1, 9, 15, 128, 128, 2000
    cls = torch.zeros(N, A * C, H, W, device='cuda')
    reg = torch.zeros(N, A * 5, H, W, device='cuda')
    anchors = torch.rand(H * W * A, 5, device='cuda') * 50 + 5
    boxes = torch.full((N, k, 5), float('nan'), device='cuda')
    scores = torch.full((N, k, C + 1), float('nan'), device='cuda')
    assert fr_boxes.level_pool(cls, reg, anchors, A, C, k, None, boxes, scores, 0) == k
    assert torch.equal(boxes[0], anchors[:k])
    k = 4096
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(H * W * A, generator=g)
    logit = (perm.float() / (H * W * A) * 8 - 4)        # distinct scores
    cls = torch.full((N, A * C, H, W), -20.0)
    # class 0 of anchor a at (h, w) carries the logit of row (h * W + w) * A + a
    cls.view(N, A, C, H, W)[0, :, 0] = logit.view(H, W, A).permute(2, 0, 1)
    cls = cls.cuda()
    boxes = torch.full((N, k, 5), float('nan'), device='cuda')
    scores = torch.full((N, k, C + 1), float('nan'), device='cuda')
    assert fr_boxes.level_pool(cls, reg, anchors, A, C, k, None, boxes, scores, 0) == k
    want = torch.topk(logit, k)[1].cuda()
    assert torch.equal(boxes[0], anchors[want])
    assert torch.allclose(scores[0, :, 0], torch.sigmoid(logit.cuda()[want]), rtol=0, atol=1e-6)


def test_levels_beyond_the_library_limits_fall_back_per_level(monkeypatch):
    """a level with more rows than the select kernel holds (1 000 000; a 4096^2 image at stride 8 has
    2.36 M) or nms_pre > 4096 used to raise; such a level now takes the op-by-op form while the other levels still
    go through the library.  The limits are lowered here instead of building a 2.36 M-row level; the library's own
    refusal at the real limit is checked through the C ABI.  nms_pre = None means no cut."""
    from r3det import _C
    from r3det.ops import fr_boxes
    head = make_head(False, (8, 16, 32))
    A, C, N = head.num_anchors, 15, 2
    sizes = [(32, 32), (12, 12), (4, 4)]
    g = torch.Generator().manual_seed(9)
    cls = [spread_logits(N, A, C, h, w, 30 + i).cuda() for i, (h, w) in enumerate(sizes)]
    reg = [(torch.randn(N, A * 5, h, w, generator=g) * 0.3).cuda() for h, w in sizes]
    cfg = dict(nms_pre=300)
    calls = []
    real = fr_boxes.level_pool
    monkeypatch.setattr(fr_boxes, "level_pool", lambda *a, **k: (calls.append(a[0].shape[-1]), real(*a, **k))[1])
    with torch.no_grad():
        wb, ws = head.decode_bboxes_torch(cls, reg, (256, 256), cfg)
        monkeypatch.setattr(fr_boxes, "POOL_MAX_ROWS", 5000)   # level 0 has 9216 rows
        gb, gs = head.decode_bboxes(cls, reg, (256, 256), cfg)
        assert calls == [12, 4]
        monkeypatch.setattr(fr_boxes, "POOL_MAX_ROWS", 1_000_000)
        monkeypatch.setattr(fr_boxes, "POOL_MAX_K", 200)       # every cut level
        kb, ks = head.decode_bboxes(cls, reg, (256, 256), cfg)
        assert calls == [12, 4, 4]
        nb, ns = head.decode_bboxes(cls, reg, (256, 256), dict(nms_pre=None))
    for b, s in ((gb, gs), (kb, ks)):
        assert torch.allclose(s, ws, rtol=0, atol=1e-6) and torch.allclose(b, wb, rtol=1e-6, atol=1e-4)
    assert nb.shape == (N, sum(h * w * A for h, w in sizes), 5) and float(ns[..., -1].abs().max()) == 0
    L = _C.lib()
    assert L.r3det_level_pool_workspace_bytes(1, 9, 512, 512, 2000) > 0  # 2.36 M rows: sized, but the call refuses:
    t = torch.zeros(16, device='cuda')
    import ctypes
    st = (ctypes.c_longlong * 4)(0, 0, 0, 0)
    assert L.r3det_level_pool(_C.ptr(t), st, _C.ptr(t), st, _C.ptr(t), 0, 1, 9, 15, 512, 512, 2000, 4.0, -1.0, -1.0,
                              _C.ptr(t), _C.ptr(t), 2000, 0, _C.ptr(t), 1 << 30, None) == -1


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("per_image", [False, True])
@pytest.mark.parametrize("nms_pre", [300, -1, 5000])
def test_levels_pool_equals_level_by_level(per_image, channels_last, nms_pre):
    """r3det_levels_pool (all levels, three launches) writes bit for bit what one r3det_level_pool call per level
    writes: levels that are cut and levels that are not, in any mix, one pool."""
    from r3det.ops import fr_boxes
    A, C, N = (1 if per_image else 3), 15, 3
    sizes = [(40, 36), (3, 5), (20, 18), (10, 9), (5, 5), (1, 1)]
    g = torch.Generator().manual_seed(7)
    cls = [spread_logits(N, A, C, h, w, 30 + i).cuda() for i, (h, w) in enumerate(sizes)]
    reg = [(torch.randn(N, A * 5, h, w, generator=g) * 0.5).cuda() for h, w in sizes]
    if channels_last:
        cls = [c.contiguous(memory_format=torch.channels_last) for c in cls]
        reg = [r.contiguous(memory_format=torch.channels_last) for r in reg]
    anchors = [(torch.rand(*((N,) if per_image else ()), h * w * A, 5, generator=g) * torch.tensor([400., 300., 60., 40., 1.])
                + torch.tensor([0., 0., 4., 4., -1.5])).cuda() for h, w in sizes]
    rows = [min(nms_pre, h * w * A) if nms_pre > 0 else h * w * A for h, w in sizes]
    n = sum(rows)
    wb, ws = torch.full((N, n, 5), -7.0).cuda(), torch.full((N, n, C + 1), -7.0).cuda()
    off = 0
    for c, r, a, k in zip(cls, reg, anchors, rows):
        assert fr_boxes.level_pool(c, r, a, A, C, nms_pre, (300, 420), wb, ws, off) == k
        off += k
    gb, gs = torch.full((N, n + 3, 5), -7.0).cuda(), torch.full((N, n + 3, C + 1), -7.0).cuda()
    assert fr_boxes.levels_pool(cls, reg, anchors, A, C, nms_pre, (300, 420), gb, gs) == n
    assert torch.equal(gb[:, :n], wb) and torch.equal(gs[:, :n], ws)
    assert bool((gb[:, n:] == -7).all()) and bool((gs[:, n:] == -7).all())  # nothing behind the pool's rows


def test_levels_pool_argument_checks():
    import ctypes
    from r3det import _C
    L = _C.lib()
    i3 = (ctypes.c_int * 3)
    assert L.r3det_levels_pool_workspace_bytes(3, 2, i3(9, 9, 9), i3(128, 64, 8), i3(128, 64, 8), 2000) == \
        L.r3det_level_pool_workspace_bytes(2, 9, 128, 128, 2000) + L.r3det_level_pool_workspace_bytes(2, 9, 64, 64, 2000)
    assert L.r3det_levels_pool_workspace_bytes(0, 2, i3(9, 9, 9), i3(8, 8, 8), i3(8, 8, 8), 2000) == 0
    assert L.r3det_levels_pool_workspace_bytes(9, 2, None, None, None, 2000) == 0  # more than 8 levels
    assert L.r3det_levels_pool(0, None, None, None, None, None, 0, 1, None, 15, None, None, -1, 1.0, -1.0, -1.0, None, None,
                               0, None, 0, None) != 0
