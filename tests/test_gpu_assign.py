"""GPU parity: fused MaxIoU assignment (r3det_rbbox_assign: no K x N matrix) against
(a) the restated assigner rules applied to the dense overlaps of the same library and
(b) a numpy restatement applied to the ORACLE's overlap matrix (small sizes).
gt_inds / argmax are integers and max_overlaps are the kernel's own IoUs: exact equality."""
import numpy as np
import pytest
import torch

from helpers import rand_boxes
from oracle import api as O

pytestmark = pytest.mark.gpu

CALCS = ['RBboxOverlaps2D_v1', 'RBboxOverlaps2D_v2', 'RBboxOverlaps2D_v3']


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make(calc, **kw):
    from r3det.core.bbox.assigners import MaxIoUAssigner
    cfg = dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0., ignore_iof_thr=-1, iou_calculator=dict(type=calc))
    cfg.update(kw)
    return MaxIoUAssigner(**cfg)


def check_same(asg, boxes, gts, labels=None):
    fused = asg.assign(boxes, gts, gt_labels=labels, with_gt_stats=True)
    dense = asg.assign_wrt_overlaps(asg.iou_calculator(gts, boxes), labels)
    assert torch.equal(fused.gt_inds, dense.gt_inds)
    assert torch.equal(fused.max_overlaps, dense.max_overlaps)
    assert torch.equal(fused.argmax_overlaps, dense.argmax_overlaps)
    assert torch.equal(fused.gt_max_overlaps, dense.gt_max_overlaps)
    assert torch.equal(fused.gt_argmax_overlaps, dense.gt_argmax_overlaps)
    if labels is not None:
        assert torch.equal(fused.labels, dense.labels)
    return fused


@pytest.fixture(params=[(0, 0), (0, 100), (3, 0)], ids=["tile-queue", "tile-queue-dense-tiles", "global-queue"])
def assign_form(request):
    """Round 5: the fused assignment runs on the matrix path's queue (stream3 -> drain with the keys as its result -> the
    low-quality sweep tile by tile); iou_qcap 100 marks every tile with a fuller wave dense (the drain enumerates and
    tests its 8192 pairs); iou_impl 3: the global pair queue of rounds 2-4."""
    from r3det import _C
    _C.set_option("iou_impl", request.param[0])
    _C.set_option("iou_qcap", request.param[1])
    yield request.param
    _C.set_option("iou_impl", 0)
    _C.set_option("iou_qcap", 0)


@pytest.mark.parametrize("calc", CALCS)
@pytest.mark.parametrize("k,n,span", [(1, 100, 150.), (5, 1000, 300.), (37, 5000, 600.), (128, 20000, 1000.)])
def test_matches_dense_rules(calc, k, n, span, assign_form):
    gts = dev(rand_boxes(k, 10 + k, span=span))
    boxes = dev(rand_boxes(n, 20 + n, span=span))
    labels = torch.randint(0, 15, (k,), device='cuda')
    for kw in (dict(), dict(min_pos_iou=0.3), dict(gt_max_assign_all=False), dict(match_low_quality=False),
               dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.1)):
        check_same(make(calc, **kw), boxes, gts, labels)


def test_assignment_shape_full_size(assign_form):
    """128 DOTA-like gts against the real 196 416-anchor grid (BASELINE training-step shape)."""
    from r3det import synthetic as syn
    anchors = syn.anchor_grid(device='cuda')
    gts = syn.dota_like_rboxes(128, 5, device='cuda')
    res = check_same(make('RBboxOverlaps2D_v1'), anchors, gts, torch.randint(0, 15, (128,), device='cuda'))
    assert (res.gt_inds > 0).sum() >= 128 * 0.6  # most gts own at least their best anchor


def test_gt_without_any_overlap_takes_every_anchor():
    """min_pos_iou = 0 (the shipped config): a gt whose best IoU is 0 satisfies `overlaps[i] == gt_max[i]`
    on EVERY anchor; later gts then re-take their own best anchors.  The fused path must reproduce it."""
    boxes = dev(rand_boxes(3000, 3, span=400.))
    gts = rand_boxes(6, 4, span=400.)
    gts[2, :2] += 5000.   # far away from every box
    gts[4, :2] -= 7000.
    gts = dev(gts)
    for kw in (dict(), dict(gt_max_assign_all=False)):
        res = check_same(make('RBboxOverlaps2D_v1', **kw), boxes, gts)
        if kw == dict():
            assert (res.gt_inds > 0).all()                       # nobody stays negative
            assert (res.gt_inds == 5).sum() > 2000               # the last zero-overlap gt (index 4) wins them
    res = check_same(make('RBboxOverlaps2D_v1', min_pos_iou=0.01), boxes, gts)
    assert (res.gt_inds == 5).sum() == 0


@pytest.mark.parametrize("calc,geom", [('RBboxOverlaps2D_v1', O.V1), ('RBboxOverlaps2D_v3', O.V3)])
def test_against_oracle_matrix(calc, geom):
    """Independent of the library's own IoU kernels: numpy rules on the oracle's matrix."""
    k, n = 9, 700
    g, b = rand_boxes(k, 77, span=260.), rand_boxes(n, 78, span=260.)
    with O.twin():
        ov = O.iou_mat(geom, g, b)
    mo, am = ov.max(0), ov.argmax(0)
    gm = ov.max(1)
    want = np.full(n, -1, dtype=np.int64)
    want[(mo >= 0) & (mo < 0.4)] = 0
    want[mo >= 0.5] = am[mo >= 0.5] + 1
    for i in range(k):
        if gm[i] >= 0.:
            want[ov[i] == gm[i]] = i + 1
    res = make(calc).assign(dev(b), dev(g))
    assert np.array_equal(res.gt_inds.cpu().numpy(), want)
    assert np.array_equal(res.max_overlaps.cpu().numpy(), mo)


def test_degenerate_inputs_take_the_dense_path():
    asg = make('RBboxOverlaps2D_v1')
    boxes = dev(rand_boxes(50, 1))
    res = asg.assign(boxes, boxes.new_zeros((0, 5)))
    assert (res.gt_inds == 0).all() and res.num_gts == 0
    res = asg.assign(boxes.new_zeros((0, 5)), boxes[:3])
    assert res.gt_inds.numel() == 0


@pytest.mark.parametrize("wrt_candidates", [True, False])
def test_ignore_regions_mark_anchors_ignored(wrt_candidates):
    """ignore_iof_thr > 0 with a non-empty ignore set (mmdet 2.19 max_iou_assigner.py, restated): anchors whose IoF
    with an ignore region exceeds the threshold become -1 unless matched low-quality; the configuration leaves the
    fused path (it has no ignore input) and the result equals the rules applied to the masked dense matrix."""
    asg = make('RBboxOverlaps2D_v1', ignore_iof_thr=0.5, ignore_wrt_candidates=wrt_candidates, match_low_quality=False)
    boxes = dev(rand_boxes(3000, 4, span=400.))
    gts = dev(rand_boxes(12, 5, span=400.))
    ign = dev(rand_boxes(6, 6, span=400., lo=60., hi=200.))
    res = asg.assign(boxes, gts, gt_bboxes_ignore=ign)
    ov = asg.iou_calculator(gts, boxes)
    iof = asg.iou_calculator(boxes, ign, mode='iof').max(1)[0] if wrt_candidates else \
        asg.iou_calculator(ign, boxes, mode='iof').max(0)[0]
    hit = iof > 0.5
    assert 10 < int(hit.sum()) < 3000
    assert bool((res.gt_inds[hit] == -1).all())
    ov[:, hit] = -1
    assert torch.equal(res.gt_inds, asg.assign_wrt_overlaps(ov).gt_inds)
    # an empty ignore set, or the threshold off: unchanged, fused
    plain = make('RBboxOverlaps2D_v1', match_low_quality=False).assign(boxes, gts)
    assert torch.equal(asg.assign(boxes, gts, gt_bboxes_ignore=ign[:0]).gt_inds, plain.gt_inds)


def test_assignment_with_prepared_anchor_grid_equals_plain():
    """MaxIoUAssigner.assign(..., shared_key=) prepares the anchor grid's columns once (r3det_iou_prepare_columns) and
    reuses them for every image / step (r3det_rbbox_assign_prepared): the same assignment, bit for bit, as the plain
    call; a different grid under another key replaces the cache."""
    from r3det import synthetic as syn
    from r3det.core.bbox.assigners import MaxIoUAssigner
    anchors = syn.anchor_grid(device='cuda')
    a = MaxIoUAssigner(0.5, 0.4, 0., iou_calculator=dict(type='RBboxOverlaps2D_v1'))
    for seed in (1, 2, 3):
        gt = syn.dota_like_rboxes(128, seed, device='cuda')
        plain = a.assign(anchors, gt, with_gt_stats=True)
        prep = a.assign(anchors.clone(), gt, with_gt_stats=True, shared_key=('grid', 1024))
        assert torch.equal(plain.gt_inds, prep.gt_inds) and torch.equal(plain.max_overlaps, prep.max_overlaps)
        assert torch.equal(plain.gt_max_overlaps, prep.gt_max_overlaps)
        assert torch.equal(plain.gt_argmax_overlaps, prep.gt_argmax_overlaps)
    assert len(a._prepared_columns) == 1
    small = anchors[:4096].contiguous()
    gt = syn.dota_like_rboxes(16, 9, device='cuda')
    assert torch.equal(a.assign(small, gt).gt_inds, a.assign(small, gt, shared_key=('grid', 64)).gt_inds)
    assert len(a._prepared_columns) == 1


def test_drain_tickets_in_the_assignment_at_512_gts():
    """512 gts against the 196 416-anchor grid: the fused assignment's drain draws its blocks by atomic tickets there
    (option iou_dyn): the same result as with the static stride, and as the dense rules on the overlap matrix."""
    from r3det import _C
    from r3det import synthetic as syn
    anchors = syn.anchor_grid(device='cuda')
    gts = syn.dota_like_rboxes(512, 6, device='cuda')
    asg = make('RBboxOverlaps2D_v1')
    dyn = check_same(asg, anchors, gts)
    _C.set_option("iou_dyn", 0)
    try:
        sta = asg.assign(anchors, gts, with_gt_stats=True)
    finally:
        _C.set_option("iou_dyn", 1)
    for a, b in ((dyn.gt_inds, sta.gt_inds), (dyn.max_overlaps, sta.max_overlaps), (dyn.argmax_overlaps, sta.argmax_overlaps),
                 (dyn.gt_max_overlaps, sta.gt_max_overlaps), (dyn.gt_argmax_overlaps, sta.gt_argmax_overlaps)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("k,n", [(3, 200), (5, 21824), (1, 196416), (2000, 100), (10000, 100)])
def test_skinny_shapes_keep_a_small_workspace_and_the_same_result(k, n):
    """a tile of the tile-queue form costs 40 KB however empty it is -- n1 = 10000, n2 = 100 asked for ~50 MB
    against the global queue's 8 MB.  The tile form is now taken only where its layout is no larger than the global
    queue's (or small in absolute terms) and the workspace size follows: never more than max(global queue, 1 MB) + the
    fixed parts; results unchanged."""
    from r3det import _C
    L = _C.lib()
    ws = int(L.r3det_rbbox_assign_workspace_bytes(k, n))
    fixed = 64 * (k + n) + 20 * (k + n) + 8 * 256          # records, keys, low-quality marks, alignment
    assert ws <= max(8 * k * n, 1 << 20) + 40960 + fixed, (ws, k, n)
    gts = dev(rand_boxes(k, 3 + k, span=400.))
    boxes = dev(rand_boxes(n, 5 + n, span=400.))
    if k * n <= 4_000_000:
        check_same(make('RBboxOverlaps2D_v1'), boxes, gts, torch.randint(0, 15, (k,), device='cuda'))


def test_labels_from_the_kernel_equal_mmdets_rule_incl_ignored_and_int32_labels():
    """assigned_labels = -1, and gt_labels[gt_inds - 1] at the positives (mmdet max_iou_assigner.py, end of
    assign_wrt_overlaps), written by the kernel that writes gt_inds (r3det_rbbox_assign_labeled)."""
    gts = dev(rand_boxes(40, 1, span=300.))
    boxes = dev(rand_boxes(6000, 2, span=300.))
    asg = make('RBboxOverlaps2D_v1', pos_iou_thr=0.6, neg_iou_thr=0.3, match_low_quality=False)   # (IoU in [0.3, 0.6): -1 = ignored)
    for labels in (torch.randint(0, 15, (40,), device='cuda'), torch.randint(0, 15, (40,), device='cuda', dtype=torch.int32)):
        res = asg.assign(boxes, gts, gt_labels=labels)
        want = torch.full_like(res.gt_inds, -1)
        pos = res.gt_inds > 0
        want[pos] = labels.long()[res.gt_inds[pos] - 1]
        assert res.labels.dtype == torch.int64 and torch.equal(res.labels, want)
        assert bool((res.gt_inds == -1).any()) and bool((res.gt_inds == 0).any()) and bool(pos.any())
    assert asg.assign(boxes, gts).labels is None
