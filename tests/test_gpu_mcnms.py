"""GPU parity: the batched multiclass NMS pipeline (r3det_mcnms_select / r3det_mcnms_v1) against
(a) outputs recorded from the reference's own multiclass_nms_rotated (tests/golden/wrappers.npz)
and (b) the per-image operator path, which the other GPU tests pin to the oracle.  Detections are
gathers of the inputs and labels are integers: everything is compared for exact equality."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu

CFG = dict(iou_thr=0.1)


def pools(B, n, seed, frac_pos=0.6, classes=15):
    from r3det import synthetic as syn
    ps = [syn.nms_pool(n, seed + 17 * i, num_classes=classes, frac_pos=frac_pos, device='cuda') for i in range(B)]
    return torch.stack([p[0] for p in ps]), torch.stack([p[1] for p in ps])


def same(batch_out, boxes, scores, thr, cfg, max_num):
    from r3det.core.post_processing import multiclass_nms_rotated
    assert len(batch_out) == boxes.size(0)
    for i, (d, lab) in enumerate(batch_out):
        rd, rl = multiclass_nms_rotated(boxes[i], scores[i], thr, cfg, max_num)
        assert d.shape == rd.shape and lab.dtype == torch.int64
        assert torch.equal(d, rd), f"image {i}: dets differ"
        assert torch.equal(lab, rl), f"image {i}: labels differ"


@pytest.fixture(params=[0, 100], ids=["queue", "overflow"])
def qcap(request):
    from r3det import _C
    _C.set_option("nms_qcap", request.param)
    yield request.param
    _C.set_option("nms_qcap", 0)


@pytest.fixture(params=['v1', 'v3', 'v2'])
def nms_type(request):
    return request.param


@pytest.mark.parametrize("max_num", [50, 2000, -1])
def test_golden_reference_wrapper(max_num, nms_type):
    """The reference's wrapper ran on these inputs (make_golden_wrappers.py); B = 2 repeats the
    image so that the image stride of every array is exercised."""
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    g = np.load(os.path.join(GOLDEN, "wrappers.npz"))
    b = torch.from_numpy(g["mc_boxes"]).cuda()
    s = torch.from_numpy(g["mc_scores"]).cuda()
    cfg = dict(type=nms_type, iou_thr=0.1)
    out = multiclass_nms_rotated_batch(torch.stack([b, b]), torch.stack([s, s]), 0.05, cfg, max_num)
    if max_num > 0:
        for d, lab in out:
            assert np.array_equal(d.cpu().numpy(), g[f"mc_{nms_type}_{max_num}_dets"])
            assert np.array_equal(lab.cpu().numpy(), g[f"mc_{nms_type}_{max_num}_labels"])
    same(out, torch.stack([b, b]), torch.stack([s, s]), 0.05, cfg, max_num)


@pytest.mark.parametrize("B,n", [(1, 100), (3, 1000), (4, 5344), (2, 9000)])
@pytest.mark.parametrize("max_num", [2000, 37])
def test_matches_per_image_path(B, n, max_num, qcap, nms_type):
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    boxes, scores = pools(B, n, 1000 + n)
    cfg = dict(type=nms_type, iou_thr=0.1)
    same(multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, max_num), boxes, scores, 0.05, cfg, max_num)


def test_ragged_and_empty_images(nms_type):
    """Images with very different candidate counts, one with none, in one batch; then all empty."""
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    CFG = dict(type=nms_type, iou_thr=0.1)
    boxes, scores = pools(4, 2000, 5)
    scores[1] = 0.01                      # no candidate at all
    scores[2, 50:] = 0.0                  # a handful
    scores[3, :, :-1] = scores[3, :, :-1].clamp(min=0.06)  # EVERY (anchor, class) pair: 30 000 candidates
    out = multiclass_nms_rotated_batch(boxes, scores, 0.05, CFG, 2000)
    assert out[1][0].shape == (0, 6) and out[1][1].shape == (0,)
    same(out, boxes, scores, 0.05, CFG, 2000)
    out = multiclass_nms_rotated_batch(boxes[:2], torch.zeros_like(scores[:2]), 0.05, CFG, 2000)
    assert all(d.shape == (0, 6) and lab.shape == (0,) and lab.dtype == torch.int64 for d, lab in out)


def test_score_ties_keep_candidate_order(nms_type):
    """Quantised scores: thousands of exact ties; the stable sort must order them like the
    per-image path (torch.sort(stable=True))."""
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    CFG = dict(type=nms_type, iou_thr=0.1)
    boxes, scores = pools(2, 3000, 77)
    scores = (scores * 8).round() / 8
    same(multiclass_nms_rotated_batch(boxes, scores, 0.05, CFG, 2000), boxes, scores, 0.05, CFG, 2000)


def test_single_class_and_other_types():
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    boxes, scores = pools(2, 800, 9, classes=1)
    same(multiclass_nms_rotated_batch(boxes, scores, 0.05, CFG, 100), boxes, scores, 0.05, CFG, 100)
    boxes, scores = pools(2, 800, 10)
    for cfg in (dict(type='v3', iou_thr=0.1), dict(type='v2', iou_thr=0.1), dict(type='mmcv', iou_thr=0.1)):
        same(multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 100), boxes, scores, 0.05, cfg, 100)


def test_v3_thin_boxes_never_kept_nor_suppress():
    """obb_nms drops boxes with min(w, h) < 1e-3 before its kernel (nms_rotated_wrapper.py:40-46):
    they are not output and cannot suppress; a thin top-score box sits on every fourth box."""
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    boxes, scores = pools(3, 1500, 31)
    boxes[:, ::4, 3] = 5e-4
    boxes[1, 1::4, 2] = 0.0
    cfg = dict(type='v3', iou_thr=0.1)
    out = multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 2000)
    for d, _ in out:
        assert d.size(0) > 0 and bool((d[:, 2:4].min(1)[0] >= 0.001).all())
    same(out, boxes, scores, 0.05, cfg, 2000)
    boxes[2, :, 2] = 0.0  # an image with candidates but no live box
    out = multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 2000)
    assert out[2][0].shape == (0, 6)
    same(out, boxes, scores, 0.05, cfg, 2000)


@pytest.mark.parametrize("max_num", [-1, 0, -3])
def test_v2_non_positive_max_num_slices_like_reference(max_num):
    """bbox_nms_rotated.py:63-65 slices [:max_num] whenever kept > max_num: -1 drops the last
    detection, 0 drops all."""
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    boxes, scores = pools(2, 600, 3)
    cfg = dict(type='v2', iou_thr=0.1)
    same(multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, max_num), boxes, scores, 0.05, cfg, max_num)


def test_mcnms_argument_errors():
    """r3det_mcnms refuses what it cannot run instead of guessing: unknown nms type, missing maxc
    for type 1, too small a workspace."""
    from r3det import _C
    L = _C.lib()
    B, n, K, cap = 1, 64, 1, 64
    dev = torch.device('cuda')
    boxes = torch.rand(B, n, 5, device=dev) * 50 + 10
    i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)  # noqa: E731
    row, lab, rank, cnt, kept = i32(B, n), i32(B, n), i32(B, n), torch.full((B,), n, dtype=torch.int32, device=dev), i32(B)
    row[0] = torch.arange(n, dtype=torch.int32, device=dev)
    sc = torch.rand(B, n, device=dev)
    mx = torch.full((B,), 60., device=dev)
    wsb = int(L.r3det_mcnms_workspace_bytes(B, cap))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dets, labels = torch.empty(B, n, 6, device=dev), torch.empty(B, n, dtype=torch.int64, device=dev)

    def call(t, maxc, ws_bytes):
        return L.r3det_mcnms(t, _C.ptr(boxes), B, n, K, _C.ptr(row), _C.ptr(lab), _C.ptr(sc), _C.ptr(rank), _C.ptr(cnt),
                             _C.ptr(maxc) if maxc is not None else None, cap, 0.1, n, _C.ptr(ws), ws_bytes,
                             _C.ptr(dets), _C.ptr(labels), None, _C.ptr(kept), _C.stream())
    assert call(0, mx, wsb) != 0 and call(4, mx, wsb) != 0
    assert call(1, None, wsb) != 0
    assert call(3, None, wsb - 1) != 0
    for t in (1, 2, 3):
        assert call(t, mx, wsb) == 0
        torch.cuda.synchronize()
        assert 0 < int(kept[0]) <= n


def test_guessed_workspace_size_is_checked_and_redone(nms_type):
    """After the first call of a shape the workspace size is guessed from the previous counts and the
    counts are read at the end: a batch with 9 x more candidates than the last one must come out right
    (the library clamps to the guessed capacity, the wrapper notices and runs again), and so must a
    much smaller one."""
    from r3det.core.post_processing import CapacityHint, multiclass_nms_rotated_batch
    cfg = dict(type=nms_type, iou_thr=0.1)
    hint = CapacityHint()
    boxes, scores = pools(2, 1200, 41)
    few = scores.clone()
    few[:, 100:] = 0.0
    many = scores.clone()
    many[:, :, :-1] = many[:, :, :-1].clamp(min=0.06)
    for s in (few, many, few, scores, many):
        same(multiclass_nms_rotated_batch(boxes, s, 0.05, cfg, 300, hint=hint), boxes, s, 0.05, cfg, 300)
    assert len(hint._last) == 1


@pytest.mark.parametrize("m_target", [65472, 65500, 65535, 65600])
def test_pool_at_the_capacity_edge(m_target):
    """candidate counts in 65473..65535 round up to cap = 65536, which the library refuses;
    the wrapper must take the per-image path from 65473 on and the batched path up to 65472, with and
    without a capacity hint.  Boxes sit on a sparse grid (no overlaps) so that the 65 k-box NMS is cheap."""
    from r3det.core.post_processing import CapacityHint, multiclass_nms_rotated_batch
    n, C = 4400, 15
    g = torch.Generator().manual_seed(m_target)
    ij = torch.arange(n)
    boxes = torch.stack([(ij % 70) * 40.0 + 5, (ij // 70) * 40.0 + 5, torch.full((n,), 12.0),
                         torch.full((n,), 8.0), -torch.rand(n, generator=g)], 1)
    scores = torch.zeros(n, C + 1)
    flat = torch.randperm(n * C, generator=g)[:m_target]
    scores.view(-1)[(flat // C) * (C + 1) + flat % C] = 0.06 + 0.9 * torch.rand(m_target, generator=g)
    boxes, scores = boxes.cuda()[None].repeat(2, 1, 1), scores.cuda()[None].repeat(2, 1, 1)
    scores[1, :, :] = 0
    scores[1, :50, 0] = 0.5
    assert int((scores[0, :, :-1] > 0.05).sum()) == m_target
    hint = CapacityHint()
    for h in (None, hint, hint):
        out = multiclass_nms_rotated_batch(boxes, scores, 0.05, CFG, 2000, hint=h)
        assert out[0][0].shape == (2000, 6) and out[1][0].shape == (50, 6)
    same(out, boxes, scores, 0.05, CFG, 2000)


def test_label_group_reducer_equals_single_workgroup(nms_type):
    """The batched pipeline reduces every (image, label mod 16) group in its own workgroup when no suppressor edge
    joins two groups; option nms_impl = 2 forces the one-workgroup-per-image reducer: identical results, also with 40
    classes (groups hold several labels) and with one class."""
    from r3det import _C
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    cfg = dict(type=nms_type, iou_thr=0.1)
    for B, n, classes in ((4, 5344, 15), (2, 3000, 40), (1, 700, 1)):
        boxes, scores = pools(B, n, 91 + classes, classes=classes)
        got = multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 2000)
        _C.set_option("nms_impl", 2)
        try:
            want = multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 2000)
        finally:
            _C.set_option("nms_impl", 0)
        for (d, lab), (wd, wl) in zip(got, want):
            assert torch.equal(d, wd) and torch.equal(lab, wl), (nms_type, B, n, classes)


@pytest.mark.parametrize("ver", ["v1", "v3"])
def test_box_across_the_class_offset_takes_the_whole_image_reducer(ver):
    """v1 / v3 separate the classes by coordinate offsets (rnms_wrapper.py:58-63, nms_rotated_wrapper.py:84-90); a box
    wide enough to reach across an offset suppresses boxes of ANOTHER class in the reference.  The drain flags such an
    edge and the reducer then runs the whole image in one workgroup: same detections as the per-image wrapper path."""
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    g = torch.Generator().manual_seed(5)
    n, C = 600, 15
    b = torch.rand(n, 5, generator=g) * torch.tensor([100., 100., 12., 12., 1.]) + torch.tensor([0., 0., 2., 2., -1.5])
    # class 0: a 100-wide box centred at x = 100; class 1: boxes at x ~ 0, which the offset (max + 1 = 101 for v1)
    # moves to x ~ 101 -- inside the wide box
    b[0] = torch.tensor([100., 50., 100., 100., 0.])
    b[1:40, 0] = torch.rand(39, generator=g) * 3
    b[1:40, 1] = 50 + torch.rand(39, generator=g) * 20
    s = torch.rand(n, C + 1, generator=g) * 0.04
    s[0, 0] = 0.99
    s[1:40, 1] = 0.5 + 0.4 * torch.rand(39, generator=g)
    idx = torch.arange(40, n)
    s[idx, torch.randint(0, C, (n - 40,), generator=g)] = 0.06 + 0.9 * torch.rand(n - 40, generator=g)
    s[:, -1] = 0
    boxes, scores = torch.stack([b, b.flip(0)]).cuda(), torch.stack([s, s.flip(0)]).cuda()
    cfg = dict(type=ver, iou_thr=0.1)
    out = multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 2000)
    same(out, boxes, scores, 0.05, cfg, 2000)
    if ver == "v1":
        # the cross-class suppression really happens: some class-1 box near x = 0 is missing from the detections
        d, lab = out[0]
        kept1 = int((lab == 1).sum())
        assert kept1 < int((s[1:40, 1] > 0.05).sum())


@pytest.mark.parametrize("piles,per_pile,classes", [(40, 40, 5), (6, 300, 3), (1, 2000, 1), (3, 700, 15), (50, 34, 1)])
def test_reducer_rows_with_long_suppressor_lists(piles, per_pile, classes, nms_type):
    """Piles of near-identical boxes: the j-th box of a pile has j suppressors -- beyond the 32-entry list the drain
    keeps them as overflow-mask bits, which the in-register reducer turns into LDS entries (40 x 40: 1120 entries),
    or, when the LDS list is full (300- and 2000-box piles: 10^4 .. 10^6 entries), leaves to the sequential tail.  One
    or two rows per thread (2000 rows in one label), deep chains.  Equal to the one-workgroup reducer and to the
    per-image wrapper path."""
    from r3det import _C
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    g = torch.Generator().manual_seed(piles * 1000 + per_pile)
    n = piles * per_pile
    centres = torch.stack([torch.arange(piles) % 9, torch.arange(piles) // 9], 1).float() * 100 + 60  # well apart
    b = torch.zeros(n, 5)
    pile = torch.arange(n) % piles
    b[:, :2] = centres[pile] + torch.randn(n, 2, generator=g) * 0.3
    b[:, 2] = 30 + torch.rand(n, generator=g)
    b[:, 3] = 12 + torch.rand(n, generator=g)
    b[:, 4] = -0.4 + 0.01 * torch.randn(n, generator=g)
    s = torch.rand(n, classes + 1, generator=g) * 0.04
    s[torch.arange(n), pile % classes] = 0.06 + 0.9 * torch.rand(n, generator=g)
    s[:, -1] = 0
    boxes, scores = torch.stack([b, b.flip(0)]).cuda(), torch.stack([s, s.flip(0)]).cuda()
    cfg = dict(type=nms_type, iou_thr=0.1)
    got = multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 2000)
    for d, lab in got:
        assert d.size(0) == piles  # the best box of every pile, nothing else
    _C.set_option("nms_impl", 2)
    try:
        want = multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 2000)
    finally:
        _C.set_option("nms_impl", 0)
    for (d, lab), (wd, wl) in zip(got, want):
        assert torch.equal(d, wd) and torch.equal(lab, wl)
    if n <= 2000:
        same(got, boxes, scores, 0.05, cfg, 2000)


# ------------------------------------------------------------------------------------------------ round 5: padded form
@pytest.mark.parametrize("max_num", [50, 2000])
def test_padded_form_equals_the_lists_and_the_reference_golden(max_num, nms_type):
    """r3det_mcnms_padded (PaddedNms): the (B, max_num + 1, 7) buffer a detector hands on -- rows, zero padding, the
    count row -- against the reference wrapper's goldens and against the list form, with NO host read inside."""
    from r3det.core.post_processing import PaddedNms, multiclass_nms_rotated_batch
    g = np.load(os.path.join(GOLDEN, "wrappers.npz"))
    b = torch.from_numpy(g["mc_boxes"]).cuda()
    s = torch.from_numpy(g["mc_scores"]).cuda()
    cfg = dict(type=nms_type, iou_thr=0.1)
    boxes, scores = torch.stack([b, b, b]).contiguous(), torch.stack([s, s, s]).contiguous()
    scores[2] = 0.0  # an image without candidates
    B, n, K = 3, b.size(0), s.size(1) - 1
    pn = PaddedNms(B, n, K, 0.05, cfg, max_num, cap=n * K, device=b.device)
    pn.out[:, :max_num].fill_(7.0)  # stale rows must not survive (the count row keeps the zeros it was allocated with)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = pn(boxes, scores)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    out = out.cpu().numpy()
    lists = multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, max_num)
    assert pn.overflow.tolist() == [0, 0, 0]
    for i, (d, lab) in enumerate(lists):
        k = d.size(0)
        assert int(pn.counts[i]) == k and out[i, max_num, 0] == k and not out[i, max_num, 1:].any()
        assert np.array_equal(out[i, :k, :6], d.cpu().numpy())
        assert np.array_equal(out[i, :k, 6], lab.cpu().numpy().astype(np.float32))
        assert not out[i, k:max_num].any()
    for i in (0, 1):
        k = int(pn.counts[i])
        assert np.array_equal(out[i, :k, :6], g[f"mc_{nms_type}_{max_num}_dets"])
        assert np.array_equal(out[i, :k, 6], g[f"mc_{nms_type}_{max_num}_labels"].astype(np.float32))
    assert int(pn.counts[2]) == 0
    for (d, lab), (d2, lab2) in zip(lists, pn.lists()):
        assert torch.equal(d, d2) and torch.equal(lab, lab2)


@pytest.mark.parametrize("front", [0, 6], ids=["counting", "sorted-chunks"])
def test_padded_form_flags_a_short_capacity_and_recovers(nms_type, front):
    """(front 6: the sorted-chunk form of ranking and pair tests forced on these small pools -- an image with more
    candidates than the capacity is its first `cap` candidates there too)"""
    from r3det import _C
    from r3det.core.post_processing import PaddedNms, multiclass_nms_rotated_batch
    _C.set_option("nms_impl", front)
    try:
        _short_capacity(nms_type, PaddedNms, multiclass_nms_rotated_batch)
    finally:
        _C.set_option("nms_impl", 0)


def _short_capacity(nms_type, PaddedNms, multiclass_nms_rotated_batch):
    boxes, scores = pools(2, 3000, 77)
    cfg = dict(type=nms_type, iou_thr=0.1)
    n_cand = (scores[..., :-1] > 0.05).flatten(1).sum(1)
    pn = PaddedNms(2, 3000, scores.size(2) - 1, 0.05, cfg, 2000, cap=int(n_cand.min()) // 2, device=boxes.device)
    pn(boxes, scores)
    pn.post_flags()
    assert pn.check()                       # both images had more candidates than the capacity
    assert pn.overflow.tolist() == [1, 1]
    while pn.cap < int(n_cand.max()):
        pn.grow()
    pn(boxes, scores)
    pn.post_flags()
    assert not pn.check() and pn.overflow.tolist() == [0, 0]
    for (d, lab), (d2, lab2) in zip(multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 2000), pn.lists()):
        assert torch.equal(d, d2) and torch.equal(lab, lab2)


# ------------------------------------------------------------------------------------------------ round 5: walk reducer
@pytest.fixture(params=[4, 0, 6], ids=["walk", "rounds", "sorted-chunks"])
def reducer(request):
    """nms_impl 4: one wavefront per (image, label group) walks its rows in score order (nms_reduce_walk_kernel, round 5:
    measured, not the default); 0: the dependency-round reducer.  Greedy NMS has one answer: both must give it.
    6 (round 6): the sorted-chunk form of the front of the pipeline -- ranks by binary search in sorted chunks, the
    pair tests over the candidates in x order with whole tiles skipped by their extents -- which pools beyond 10 240
    candidates take by themselves, forced here on small ones: same records, same queue contents, same answer."""
    from r3det import _C
    _C.set_option("nms_impl", request.param)
    yield request.param
    _C.set_option("nms_impl", 0)


def test_reducers_on_chains_clusters_and_ragged_batches(reducer, nms_type):
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    cfg = dict(type=nms_type, iou_thr=0.1)
    g = torch.Generator(device='cuda').manual_seed(5)
    n, K = 1500, 15
    boxes = torch.zeros(3, n, 5, device='cuda')
    scores = torch.zeros(3, n, K + 1, device='cuda')
    # image 0: per class a line of boxes where box k overlaps only its neighbours, scores falling along the line: the
    # answer alternates kept / removed, the dependency chain is 100 long (chains across blocks of 64 rows)
    k = torch.arange(n, device='cuda')
    cls = k % K
    pos = k // K
    boxes[0, :, 0] = 10.0 + 6.0 * pos
    boxes[0, :, 1] = 50.0 + 40.0 * cls
    boxes[0, :, 2], boxes[0, :, 3] = 10.0, 20.0
    scores[0, k, cls] = 0.9 - 0.008 * pos.float() + 0.0004 * cls.float()
    # image 1: clusters of ~150 near-duplicates (rows with far more than 32 suppressors: the overflow rows of the mask)
    centre = torch.rand(10, 2, device='cuda', generator=g) * 800 + 100
    which = torch.randint(0, 10, (n,), device='cuda', generator=g)
    boxes[1, :, :2] = centre[which] + torch.randn(n, 2, device='cuda', generator=g) * 2.0
    boxes[1, :, 2] = 60 + torch.rand(n, device='cuda', generator=g) * 5
    boxes[1, :, 3] = 30 + torch.rand(n, device='cuda', generator=g) * 5
    boxes[1, :, 4] = -0.3 + torch.rand(n, device='cuda', generator=g) * 0.05
    scores[1, k, which % K] = 0.1 + 0.8 * torch.rand(n, device='cuda', generator=g)
    # image 2: a handful of candidates only
    b2, s2 = pools(1, n, 99)
    boxes[2], scores[2] = b2[0], s2[0]
    scores[2, 40:] = 0.0
    out = multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 2000)
    assert out[0][0].size(0) == K * 50 and 10 <= out[1][0].size(0) <= 40
    same(out, boxes, scores, 0.05, cfg, 2000)


def test_reducers_agree_on_random_pools(reducer, nms_type, qcap):
    from r3det.core.post_processing import multiclass_nms_rotated_batch
    boxes, scores = pools(4, 5344, 4242)
    cfg = dict(type=nms_type, iou_thr=0.1)
    same(multiclass_nms_rotated_batch(boxes, scores, 0.05, cfg, 2000), boxes, scores, 0.05, cfg, 2000)
