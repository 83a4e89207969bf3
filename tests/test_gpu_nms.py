"""GPU parity: rotated NMS (v1 / v2 / v3 / mmcv) through the C ABI vs the oracle.
Keep indices are compared for exact equality (north_star: bit-exact keep)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, rand_boxes
from oracle import api as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def case(n, seed, span):
    b = rand_boxes(n, seed, span=span)
    r = np.random.default_rng(seed + 1)
    return b, r.uniform(0.05, 1, n).astype(np.float32), r.integers(0, 15, n)


@pytest.fixture(params=["queue", "tiles", "overflow"])
def nms_impl(request):
    """queue = stream / drain / dependency-round reduce pipeline (default); tiles = in-place tile mask +
    dense reduce; overflow = queue pipeline with 100-entry queue regions: a tile that does not fit goes to the
    redo-tile list and is enumerated pair by pair by the drain (nothing is clipped inside the stream kernel)."""
    from r3det import _C
    _C.set_option("nms_impl", 1 if request.param == "tiles" else 0)
    _C.set_option("nms_qcap", 100 if request.param == "overflow" else 0)
    yield request.param
    _C.set_option("nms_impl", 0)
    _C.set_option("nms_qcap", 0)


def test_dense_single_class(nms_impl):
    """Crowded single-class pools: long suppression lists (> 16 non-zero words per row)."""
    from r3det.ops import obb_nms, rnms
    for n, span in [(3000, 150.), (4500, 400.)]:
        b = rand_boxes(n, 500 + n, span=span)
        s = np.random.default_rng(n).uniform(0.05, 1, n).astype(np.float32)
        d6 = np.hstack([b, s[:, None]])
        for thr in (0.1, 0.7):
            with O.twin():
                w1 = O.nms(O.V1, b, s, thr, strict=True, ascending=True)
                w3 = O.nms(O.V3, b, s, thr, strict=True)
            assert np.array_equal(rnms(dev(d6), thr)[1].cpu().numpy(), w1)
            assert np.array_equal(obb_nms(dev(d6), thr)[1].cpu().numpy(), w3)


@pytest.mark.parametrize("n,span", [(1, 100.), (63, 120.), (64, 120.), (65, 120.), (129, 150.),
                                    (1000, 400.), (2000, 600.), (5344, 1000.)])
@pytest.mark.parametrize("thr", [0.1, 0.5])
def test_keep_exact_vs_twin(n, span, thr, nms_impl):
    from r3det.ops import ml_nms_rotated, obb_nms, rnms
    from r3det.ops.mmcv_ops import nms_rotated
    b, s, lab = case(n, 300 + n, span)
    d6 = np.hstack([b, s[:, None]])
    with O.twin():
        w1 = O.nms(O.V1, b, s, thr, strict=True, ascending=True)
        w3 = O.nms(O.V3, b, s, thr, strict=True)
        bl = np.hstack([b, lab[:, None].astype(np.float32)])
        w2 = O.nms(O.V2, bl, s, thr, strict=True, with_label=True)
        w2n = O.nms(O.V2, b, s, thr, strict=True)
    dets, k1 = rnms(dev(d6), thr)
    assert np.array_equal(k1.cpu().numpy(), w1)
    assert np.array_equal(dets.cpu().numpy(), d6[w1])
    dets, k3 = obb_nms(dev(d6), thr)
    assert np.array_equal(k3.cpu().numpy(), w3)
    k2 = ml_nms_rotated(dev(b), dev(s), dev(lab), thr)
    assert np.array_equal(k2.cpu().numpy(), w2)
    from r3det.ops import nms as nms_mod
    # mmcv stand-in: labels are carried but (as recalled from mmcv 1.3.15..1.5.0) not compared
    dm, km = nms_rotated(dev(b), dev(s), thr, dev(lab))
    assert np.array_equal(km.cpu().numpy(), w2n)
    assert np.array_equal(dm.cpu().numpy(), np.hstack([b[w2n], s[w2n, None]]))
    dm, km = nms_rotated(dev(b), dev(s), thr)
    assert np.array_equal(km.cpu().numpy(), w2n)
    nms_mod.MMCV_LABEL_GUARD = True  # opt-in: the ml_nms_rotated guard
    try:
        dm, km = nms_rotated(dev(b), dev(s), thr, dev(lab))
    finally:
        nms_mod.MMCV_LABEL_GUARD = False
    assert np.array_equal(km.cpu().numpy(), w2)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 500, 2000, 8576])
def test_keep_vs_reference_golden(n):
    """Golden keep lists come from the reference CPU code (>= threshold, libm trig).  The HIP
    path uses > and the trig twin; a keep list can only differ if some decisive IoU lies within
    float rounding of thr, which the seeded fixtures do not contain."""
    from r3det.ops import ml_nms_rotated, obb_nms, rnms
    g = np.load(os.path.join(GOLDEN, "nms.npz"))
    b, s, lab = g[f"boxes_{n}"], g[f"scores_{n}"], g[f"labels_{n}"]
    d6 = np.hstack([b, s[:, None]])
    for thr in (0.1, 0.5):
        tag = f"{n}_{int(thr * 100):02d}"
        assert np.array_equal(rnms(dev(d6), thr)[1].cpu().numpy(), g[f"v1_{tag}"])
        assert np.array_equal(obb_nms(dev(d6), thr)[1].cpu().numpy(), g[f"v3_{tag}"])
        if f"v2_{tag}" in g:
            k = ml_nms_rotated(dev(b), dev(s), dev(lab.astype(np.int64)), thr)
            assert np.array_equal(k.cpu().numpy(), g[f"v2_{tag}"])


def test_idempotent_and_sorted_full_size():
    """n = 8576 (RRetinaNet pool): NMS of the kept set keeps everything; v1 keep is ascending,
    v3 keep is score-descending; kept boxes are pairwise below the threshold."""
    from r3det.ops import obb_nms, rbbox_iou, rnms
    b, s, _ = case(8576, 77, 1500.)
    d6 = dev(np.hstack([b, s[:, None]]))
    dets, keep = rnms(d6, 0.1)
    k = keep.cpu().numpy()
    assert (np.diff(k) > 0).all() and 0 < len(k) < 8576
    dets2, keep2 = rnms(dets.contiguous(), 0.1)
    assert len(keep2) == len(keep)
    iou = rbbox_iou(dets[:, :5].contiguous(), dets[:, :5].contiguous())
    iou.fill_diagonal_(0)
    # greedy order: a kept box may exceed thr only against a LOWER-scored ... no: all kept pairs
    # are mutually unsuppressed in score order, i.e. IoU(high, low) <= thr
    sc = dets[:, 5]
    hi_first = sc[:, None] > sc[None, :]
    assert float((iou * hi_first).max()) <= 0.1
    dets3, keep3 = obb_nms(d6, 0.1)
    assert (np.diff(dets3[:, 5].cpu().numpy()) <= 0).all()


def test_threshold_is_strict_greater():
    from r3det.ops import obb_nms, rnms
    b = np.array([[50, 50, 20, 10, 0, 0.9], [60, 50, 20, 10, 0, 0.8]], np.float32)
    thr = float(np.float32(0.3333333432674408))
    assert rnms(dev(b), thr)[1].tolist() == [0, 1]          # IoU == thr: CUDA keeps both (>)
    assert rnms(dev(b), np.nextafter(np.float32(thr), np.float32(0)))[1].tolist() == [0]
    assert obb_nms(dev(b), thr)[1].tolist() == [0, 1]
    # ties: lowest index wins (stable sort)
    t = np.array([[50, 50, 20, 10, 0, .5], [500, 500, 20, 10, 0, .5], [52, 50, 20, 10, 0, .5]], np.float32)
    assert rnms(dev(t), 0.1)[1].tolist() == [0, 1]
    assert obb_nms(dev(t), 0.1)[1].tolist() == [0, 1]


def test_empty_and_small_boxes():
    from r3det.ops import batched_rnms, obb_nms, rnms
    e = torch.zeros(0, 6, device='cuda')
    d, k = rnms(e, 0.1)
    assert d.shape == (0, 6) and k.numel() == 0 and k.dtype == torch.long
    d, k = obb_nms(e, 0.1)
    assert d.shape == (0, 6) and k.numel() == 0
    # obb_nms drops boxes thinner than 1e-3 before the kernel and maps indices back
    b = np.array([[50, 50, 5e-4, 10, 0, 0.99], [50, 50, 20, 10, 0, 0.9], [51, 50, 20, 10, 0, 0.8],
                  [300, 300, 20, 1e-4, 0, 0.7], [300, 300, 20, 10, 0, 0.6]], np.float32)
    d, k = obb_nms(dev(b), 0.1)
    assert k.tolist() == [1, 4]
    allsmall = b.copy()
    allsmall[:, 2] = 1e-4
    assert obb_nms(dev(allsmall), 0.1)[1].numel() == 0
    # numpy round trip
    d, k = rnms(b, 0.1, device_id=0)
    assert isinstance(k, np.ndarray) and isinstance(d, np.ndarray)


def np_multiclass(version, mb, ms, score_thr, iou_thr, max_num):
    """numpy + oracle restatement of multiclass_nms_rotated's non-mmcv branch
    (bbox_nms_rotated.py:97-131) and its batched helpers, for the wrapper-level checks."""
    C = ms.shape[1] - 1
    scores = ms[:, :-1]
    valid = scores > score_thr
    idx = np.argwhere(valid)
    boxes = mb[idx[:, 0]]
    sc = scores[valid]
    labels = idx[:, 1]
    if len(boxes) == 0:
        return np.zeros((0, 6), np.float32), np.zeros((0,), np.int64)
    if version == 'v1':
        off = (labels.astype(np.float32) * (boxes.max() + np.float32(1))).astype(np.float32)
        sh = boxes.copy()
        sh[:, :2] += off[:, None]
        with O.twin():
            keep = O.nms(O.V1, sh, sc, iou_thr, strict=True, ascending=True)
    elif version == 'v3':
        c, s = np.cos(boxes[:, 4]), np.sin(boxes[:, 4])
        xb = np.abs(boxes[:, 2] / 2 * c) + np.abs(boxes[:, 3] / 2 * s)
        yb = np.abs(boxes[:, 2] / 2 * s) + np.abs(boxes[:, 3] / 2 * c)
        hbb = np.stack([boxes[:, 0] - xb, boxes[:, 1] - yb, boxes[:, 0] + xb, boxes[:, 1] + yb], 1)
        off = labels.astype(np.float32) * (hbb.max() - hbb.min() + np.float32(1))
        sh = boxes.copy()
        sh[:, :2] = sh[:, :2] + off[:, None].astype(np.float32)
        with O.twin():
            keep = O.nms(O.V3, sh, sc, iou_thr, strict=True)
    elif version == 'mmcv':  # class-agnostic v2 geometry (ops/nms.py: MMCV_LABEL_GUARD)
        with O.twin():
            keep = O.nms(O.V2, boxes, sc, iou_thr, strict=True)
        return np.hstack([boxes[keep], sc[keep, None]]), labels[keep]
    else:
        bl = np.hstack([boxes, labels[:, None].astype(np.float32)])
        with O.twin():
            keep = O.nms(O.V2, bl, sc, iou_thr, strict=True, with_label=True)
        if len(keep) > max_num:
            keep = keep[np.argsort(-sc[keep], kind="stable")[:max_num]]
        return np.hstack([boxes[keep], sc[keep, None]]), labels[keep]
    if max_num > 0:
        keep = keep[:max_num]
    return np.hstack([boxes[keep], sc[keep, None]]), labels[keep]


@pytest.mark.parametrize("version", ['v1', 'v2', 'v3', None])
@pytest.mark.parametrize("max_num", [50, 2000])
def test_multiclass_nms_rotated(version, max_num):
    from r3det.core.post_processing import multiclass_nms_rotated
    n, C = 1500, 15
    mb = rand_boxes(n, 91, span=500.)
    r = np.random.default_rng(92)
    ms = (r.uniform(0, 1, (n, C + 1)) ** 12).astype(np.float32)  # ~18 % of (box, class) > 0.05... sparse
    cfg = dict(iou_thr=0.1) if version is None else dict(type=version, iou_thr=0.1)
    dets, labels = multiclass_nms_rotated(dev(mb), dev(ms), 0.05, cfg, max_num)
    wd, wl = np_multiclass(version or 'v1', mb, ms, 0.05, 0.1, max_num)
    assert np.array_equal(labels.cpu().numpy(), wl)
    assert np.array_equal(dets.cpu().numpy(), wd)
    assert len(wd) <= max_num
    if (version or 'v1') == 'v1' and max_num == 50:
        assert not (np.diff(wd[:, 5]) <= 0).all()  # v1 truncates in anchor order, not by score
    # nothing above threshold -> (0, 6), (0,)
    d0, l0 = multiclass_nms_rotated(dev(mb), dev(ms * 0), 0.05, cfg, max_num)
    assert d0.shape == (0, 6) and l0.shape == (0,) and l0.dtype == torch.long


def test_multiclass_mmcv_branch():
    from r3det.core.post_processing import multiclass_nms_rotated
    n, C = 800, 15
    mb = rand_boxes(n, 93, span=400.)
    ms = (np.random.default_rng(94).uniform(0, 1, (n, C + 1)) ** 12).astype(np.float32)
    dets, labels, inds = multiclass_nms_rotated(dev(mb), dev(ms), 0.05, dict(type='mmcv', iou_thr=0.1), 100,
                                                return_inds=True)
    wd, wl = np_multiclass('mmcv', mb, ms, 0.05, 0.1, 10 ** 9)
    assert np.array_equal(dets.cpu().numpy(), wd[:100])
    assert np.array_equal(labels.cpu().numpy(), wl[:100])
    assert (np.diff(dets[:, 5].cpu().numpy()) <= 0).all()


@pytest.mark.parametrize("n", [70, 500, 3000])
def test_long_suppression_chain(n):
    """A line of boxes in which box k overlaps only its neighbours and the scores decrease along the line:
    the greedy answer alternates kept / removed and its dependency chain is n long -- beyond the reducer's
    round budget (R_MAX_ROUNDS = 32), so the in-order finish of nms_reduce_rounds_kernel decides the rest."""
    from r3det.ops import ml_nms_rotated, obb_nms, rnms
    b = np.zeros((n, 5), np.float32)
    b[:, 0] = 10.0 + 6.0 * np.arange(n)  # 10 wide, 6 apart: IoU(k, k+1) = 4/16 = 0.25, IoU(k, k+2) = 0
    b[:, 1] = 50.0
    b[:, 2] = 10.0
    b[:, 3] = 20.0
    s = np.linspace(0.9, 0.1, n).astype(np.float32)
    perm = np.random.default_rng(n).permutation(n)  # original order shuffled: keep is reported in original indices
    b, s = b[perm], s[perm]
    d6 = np.hstack([b, s[:, None]])
    with O.twin():
        w1 = O.nms(O.V1, b, s, 0.1, strict=True, ascending=True)
        w3 = O.nms(O.V3, b, s, 0.1, strict=True)
    assert len(w1) == (n + 1) // 2
    assert np.array_equal(rnms(dev(d6), 0.1)[1].cpu().numpy(), w1)
    assert np.array_equal(obb_nms(dev(d6), 0.1)[1].cpu().numpy(), w3)
    k2 = ml_nms_rotated(dev(b), dev(s), dev(np.zeros(n, np.int64)), 0.1)
    assert np.array_equal(k2.cpu().numpy(), w3)


def test_row_with_more_in_edge_words_than_the_list_holds():
    """2 100 near-duplicates of one box: the last ones are suppressed from > 16 different 64-box words (the
    per-row word list holds 16), so the reducer scans their mask rows; plus a second far cluster."""
    from r3det.ops import rnms
    r = np.random.default_rng(5)
    n = 2100
    b = np.tile(np.array([[200., 200., 60., 30., -0.4]], np.float32), (n, 1))
    b[:, :2] += r.normal(0, 0.5, (n, 2)).astype(np.float32)
    far = np.tile(np.array([[900., 700., 40., 40., -1.0]], np.float32), (300, 1))
    far[:, :2] += r.normal(0, 0.3, (300, 2)).astype(np.float32)
    b = np.vstack([b, far])
    s = r.uniform(0.1, 1.0, len(b)).astype(np.float32)
    d6 = np.hstack([b, s[:, None]])
    with O.twin():
        want = O.nms(O.V1, b, s, 0.1, strict=True, ascending=True)
    assert len(want) == 2
    assert np.array_equal(rnms(dev(d6), 0.1)[1].cpu().numpy(), want)


@pytest.mark.parametrize("fn_name", ["batched_rnms", "obb_batched_nms"])
@pytest.mark.parametrize("class_agnostic", [False, True])
def test_batched_wrappers_one_call_equals_op_by_op(fn_name, class_agnostic, monkeypatch):
    """r3det_batched_rnms / r3det_obb_batched_nms (the wrapper's whole body in one library call) against the
    op-by-op form of the same wrapper (max / offsets / cat / sort / nms / index), incl. boxes thinner than 1e-3
    (dropped by v3) and score ties."""
    import r3det.ops.nms as M
    from r3det import synthetic as syn
    for n, seed in ((1, 3), (65, 4), (700, 5), (3000, 6)):
        b = syn.rand_rboxes(n, seed, device='cuda')
        g = torch.Generator().manual_seed(seed)
        s = torch.rand(n, generator=g).cuda()
        s[::7] = float(s[0])  # ties
        lab = torch.randint(0, 15, (n,), generator=g).cuda()
        if n > 10:
            b[3, 2] = 5e-4   # thin boxes
            b[9, 3] = 0.0
        fn = getattr(M, fn_name)
        fast_d, fast_k = fn(b, s, lab, 0.1, class_agnostic=class_agnostic)
        with monkeypatch.context() as mp:
            mp.setattr(M, "_batched_rnms_device", lambda *a, **k: None)
            slow_d, slow_k = fn(b, s, lab, 0.1, class_agnostic=class_agnostic)
        assert torch.equal(fast_k, slow_k), (fn_name, n)
        assert torch.equal(fast_d, slow_d), (fn_name, n)


def test_one_call_wrapper_large_pool():
    """50 000 boxes: the reducer's row-state / bit-set / worklist LDS exceeds the default 64 KB dynamic limit (raised
    with hipFuncSetAttribute); result against the op-by-op wrapper."""
    import r3det.ops.nms as M
    from r3det import synthetic as syn
    n = 50000
    b = syn.rand_rboxes(n, 3, device='cuda')
    b[:, :2] *= 6
    g = torch.Generator().manual_seed(1)
    s = torch.rand(n, generator=g).cuda()
    lab = torch.randint(0, 15, (n,), generator=g).cuda()
    assert M._batched_rnms_device(b, s, lab, 0.1, False) is None  # beyond FAST_MAX_N the wrapper goes op by op ...
    orig_max = M.FAST_MAX_N
    M.FAST_MAX_N = 65472                                          # ... the library itself takes up to 65 472 rows
    try:
        fd, fk = M._batched_rnms_device(b, s, lab, 0.1, False)
    finally:
        M.FAST_MAX_N = orig_max
    orig = M._batched_rnms_device
    M._batched_rnms_device = lambda *a, **k: None
    try:
        sd, sk = M.batched_rnms(b, s, lab, 0.1)
    finally:
        M._batched_rnms_device = orig
    assert fk.numel() > 1000 and torch.equal(fk, sk) and torch.equal(fd, sd)
    # float ``inds`` (the reference multiplies them as floats) never take the one-call form
    assert M._batched_rnms_device(b[:100], s[:100], lab[:100].float(), 0.1, False) is None
    d1, k1 = M.batched_rnms(b[:100], s[:100], lab[:100].float(), 0.1)
    d2, k2 = M.batched_rnms(b[:100], s[:100], lab[:100], 0.1)
    assert torch.equal(k1, k2) and torch.equal(d1, d2)


def test_one_call_form_at_32768_rows_exact_vs_the_twin_oracle():
    """SURVEY 8d names n = 32 768 (the reference reduces it on the host: nms_rotated_cuda.cu:117-128; rnms_kernel.cu:229-335).
    Round 6: pools up to 32 768 rows take the one-call form (FAST_MAX_N), whose ranking loop runs with four candidates
    per thread beyond 16 384 (mc_sort_prepare_kernel<., 4>): the keep list equals the twin oracle's on the offset boxes
    (ties in the scores included) and the op-by-op wrapper's."""
    import r3det.ops.nms as M
    n = 32768
    assert M.FAST_MAX_N >= n
    b = rand_boxes(n, 3, span=1500.0)
    r = np.random.default_rng(1)
    s = r.uniform(0.05, 1, n).astype(np.float32)
    s[::11] = s[5]                                  # ties: the stable order decides
    lab = r.integers(0, 15, n)
    shifted = b.copy()
    shifted[:, :2] += (lab * (b.max() + 1)).astype(np.float32)[:, None]
    with O.twin():
        want = O.nms(O.V1, shifted, s, 0.1, strict=True, ascending=True)
    tb, ts, tl = (torch.from_numpy(x).cuda() for x in (b, s, lab))
    fast = M._batched_rnms_device(tb, ts, tl, 0.1, False)
    assert fast is not None
    dets, keep = fast
    assert 5000 < keep.numel() < 20000
    assert np.array_equal(keep.cpu().numpy(), want)
    assert torch.equal(dets[:, :5], tb[keep]) and torch.equal(dets[:, 5], ts[keep])


@pytest.mark.parametrize("version", ["v1", "v3"])
@pytest.mark.parametrize("n", [100, 1030, 5000, 13000])
def test_sorted_chunk_form_equals_the_counting_form(version, n):
    """Round 6: beyond 10 240 candidates the batched pipeline ranks by binary search in sorted 1024-chunks and tests pairs
    over the candidates in x order (mc_chunk_sort_kernel, mc_sort_prepare_p_kernel, nms_stream_kernel<.., true>); option
    nms_impl 6 forces that form, 7 forbids it.  Ties in the scores, labels whose offsets leave the classes touching
    (boxes wider than the image), a count that is no multiple of 64 or 1024: keep lists and rows identical."""
    import r3det.ops.nms as M
    from r3det import _C
    b = rand_boxes(n, 5 + n, span=900.0)
    b[::97, 2] = 2500.0                               # a few boxes wider than the class offset: edges between classes
    r = np.random.default_rng(n)
    s = r.uniform(0.05, 1, n).astype(np.float32)
    s[::7] = s[3]
    lab = r.integers(0, 15, n)
    tb, ts, tl = (torch.from_numpy(x).cuda() for x in (b, s, lab))
    entry = "r3det_batched_rnms" if version == "v1" else "r3det_obb_batched_nms"
    got = {}
    try:
        for impl in (7, 6):
            _C.set_option("nms_impl", impl)
            d, k = M._batched_rnms_device(tb, ts, tl, 0.1, False, entry=entry)
            got[impl] = (d.clone(), k.clone())
    finally:
        _C.set_option("nms_impl", 0)
    assert 0 < got[7][1].numel() < n
    assert torch.equal(got[6][1], got[7][1]) and torch.equal(got[6][0], got[7][0])
    d0, k0 = M._batched_rnms_device(tb, ts, tl, 0.1, False, entry=entry)       # the default for this size
    assert torch.equal(k0, got[7][1]) and torch.equal(d0, got[7][0])


def test_sorted_chunk_form_on_a_dense_cluster_takes_the_redo_tiles():
    """3000 near-copies of one box: every tile's queue segment overflows, the tiles go to the redo list as chunks of the
    x-ordered permutation and the drain enumerates them through it."""
    import r3det.ops.nms as M
    from r3det import _C
    n = 3000
    r = np.random.default_rng(4)
    b = np.tile(np.array([[300., 300., 80., 40., 0.3]], np.float32), (n, 1))
    b[:, :2] += r.normal(0, 25.0, (n, 2)).astype(np.float32)
    b[:, 4] += r.normal(0, 0.2, n).astype(np.float32)
    s = r.uniform(0.05, 1, n).astype(np.float32)
    lab = r.integers(0, 2, n)
    tb, ts, tl = (torch.from_numpy(x).cuda() for x in (b, s, lab))
    got = {}
    try:
        for impl in (7, 6):
            _C.set_option("nms_impl", impl)
            d, k = M._batched_rnms_device(tb, ts, tl, 0.1, False)
            got[impl] = (d.clone(), k.clone())
    finally:
        _C.set_option("nms_impl", 0)
    assert 0 < got[7][1].numel() < 200
    assert torch.equal(got[6][1], got[7][1]) and torch.equal(got[6][0], got[7][0])
    # ... and the same answer in every call: every row of the pile has hundreds of suppressors, two rows per thread of
    # the label-group reducer; a count of undecided THREADS where the wavefronts' pass subtracted decided ROWS ended the
    # rounds early in one call of ten (round 6)
    for _ in range(40):
        d, k = M._batched_rnms_device(tb, ts, tl, 0.1, False)
        assert torch.equal(k, got[7][1])
    shifted = b.copy()
    shifted[:, :2] += (lab * (b.max() + 1)).astype(np.float32)[:, None]
    with O.twin():
        want = O.nms(O.V1, shifted, s, 0.1, strict=True, ascending=True)
    assert np.array_equal(got[6][1].cpu().numpy(), want)


def test_sorted_chunk_form_fuzz():
    """tools/nms_fuzz.py, 40 random pools (sizes 1 .. 20 000, 1 .. 40 classes, spread / clustered / piled / oversized
    boxes, NaN / inf / zero entries, tied scores), v1 and v3 entries: nms_impl 6 and 7 give identical rows."""
    import os
    import runpy
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "nms_fuzz.py")
    argv = sys.argv
    sys.argv = [tool, "40", "5"]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(tool, run_name="__main__")
        assert e.value.code == 0
    finally:
        sys.argv = argv


@pytest.mark.parametrize("version", ["v1", "v3"])
def test_padded_one_pool_form_equals_the_list_form(version):
    """batched_rnms_padded: the one library call of batched_rnms / obb_batched_nms without the host read of the count --
    rows [:kept] are the list form's return values (rnms_wrapper.py:34-69, nms_rotated_wrapper.py:78-98)."""
    import r3det.ops.nms as M
    from r3det.ops import batched_rnms, obb_batched_nms
    n = 3000
    b = torch.from_numpy(rand_boxes(n, 7, span=700.0)).cuda()
    g = torch.Generator().manual_seed(2)
    s = torch.rand(n, generator=g).cuda()
    lab = torch.randint(0, 15, (n,), generator=g).cuda()
    dets, keep, kept = M.batched_rnms_padded(b, s, lab, 0.1, version=version)
    want_d, want_k = (batched_rnms if version == "v1" else obb_batched_nms)(b, s, lab, 0.1)
    k = int(kept.item())
    assert k == want_k.numel() and dets.shape == (n, 6) and keep.shape == (n,)
    assert torch.equal(keep[:k], want_k) and torch.equal(dets[:k], want_d)
