"""GPU parity: rotated IoU through the C ABI (via the r3det.ops mirror) vs the oracle.

Bars: bit-exact against the oracle in twin mode (same deterministic trig, device-branch hull
sort); <= 1e-5 absolute against the golden vectors produced by the reference CPU code
(north_star tolerance for IoU)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, anchor_grid, dota_like_gt, rand_boxes
from oracle import api as O

pytestmark = pytest.mark.gpu
TOL = 1e-5  # BASELINE.json north_star: "within 1e-5 on IoU"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run(geom, a, b, iof=False):
    from r3det.ops import rbbox_iou
    from r3det.ops.mmcv_ops import box_iou_rotated
    from r3det.ops.iou import box_iou_rotated_v3
    if geom == O.V1:
        return rbbox_iou(dev(a), dev(b), False, iof).cpu().numpy()
    if geom == O.V2:
        return box_iou_rotated(dev(a), dev(b), 'iof' if iof else 'iou').cpu().numpy()
    return box_iou_rotated_v3(dev(a), dev(b), not iof).cpu().numpy()


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.fixture(params=[0, 1, 2, 4, 5], ids=["auto", "simple", "compact", "pipeline", "pipeline-overflow"])
def iou_impl(request):
    """0 = automatic (one-launch narrow-tile kernel up to 512 columns, stream + drain pipeline beyond),
    1 = one thread per pair, 2 = the wide one-launch kernel with an in-workgroup queue (the no-workspace path),
    4 = the pipeline whatever the size, 5 = the pipeline with every wave's LDS segment capped at 100 entries: a
    tile with a fuller wave is marked dense and the drain tests all of its 8192 pairs with the exact records
    (nothing is clipped inside the stream kernel)."""
    from r3det import _C
    _C.set_option("iou_impl", 4 if request.param == 5 else request.param)
    _C.set_option("iou_qcap", 100 if request.param == 5 else 0)
    yield request.param
    _C.set_option("iou_impl", 0)
    _C.set_option("iou_qcap", 0)


@pytest.mark.parametrize("geom", [O.V1, O.V2, O.V3])
@pytest.mark.parametrize("iof", [False, True])
def test_mat_bit_exact_vs_twin_dense(geom, iof, iou_impl):
    a = rand_boxes(700, 11, span=220.0, amin=-np.pi, amax=np.pi)
    b = rand_boxes(333, 12, span=220.0)
    with O.twin():
        want = O.iou_mat(geom, a, b, iof=iof, threads=8)
    got = run(geom, a, b, iof)
    assert (want > 0).mean() > 0.2
    assert same(got, want)


@pytest.mark.parametrize("geom", [O.V1, O.V2, O.V3])
def test_mat_golden_config1(geom):
    """BASELINE config 1: 1000 anchors x 128 GT."""
    g = np.load(os.path.join(GOLDEN, "iou_random.npz"))
    key = {O.V1: "v1", O.V2: "v2", O.V3: "v3"}[geom]
    got = run(geom, g["anchors"], g["gts"])
    assert np.abs(got - g[f"{key}_iou"]).max() <= TOL
    if geom != O.V2:
        assert np.abs(run(geom, g["anchors"], g["gts"], True) - g[f"{key}_iof"]).max() <= TOL
    got = run(geom, g["dense_a"], g["dense_g"])
    assert np.abs(got - g[f"dense_{key}_iou"]).max() <= TOL
    if geom == O.V1:
        assert np.abs(run(geom, g["gts"], g["anchors"]) - g["v1_iou_t"]).max() <= TOL


@pytest.mark.parametrize("geom", [O.V1, O.V2, O.V3])
def test_mat_degenerate(geom):
    g = np.load(os.path.join(GOLDEN, "iou_degenerate.npz"))
    d = g["boxes"]
    key = {O.V1: "v1", O.V2: "v2", O.V3: "v3"}[geom]
    for iof in ([False, True] if geom != O.V2 else [False]):
        with O.twin():
            want = O.iou_mat(geom, d, d, iof=iof)
        got = run(geom, d, d, iof)
        assert same(got, want)
        ref = g[f"{key}_{'iof' if iof else 'iou'}"]
        ok = (ref != -2.0) & ~np.isnan(ref)
        # boxes 30/31 carry |theta| = 7 / 100 rad: still inside the trig twin's accurate range
        assert np.abs(got[ok] - ref[ok]).max() <= TOL
        assert same(np.isnan(got[ref != -2.0]), np.isnan(ref[ref != -2.0]))


def test_odd_shapes_and_unaligned(iou_impl):
    """n2 not a multiple of 4 (scalar-store path), single row / column, > 1 column tile."""
    for (m, n) in [(1, 1), (3, 1027), (37, 2051), (130, 5)]:
        a, b = rand_boxes(m, 51 + m, span=90.0), rand_boxes(n, 52 + n, span=90.0)
        with O.twin():
            want = O.iou_mat(O.V1, a, b, threads=8)
        assert same(run(O.V1, a, b), want), (m, n)
        with O.twin():
            want = O.iou_mat(O.V3, a, b, threads=8)
        assert same(run(O.V3, a, b), want), (m, n)


@pytest.mark.parametrize("geom", [O.V1, O.V3])
@pytest.mark.parametrize("qcap", [0, 100], ids=["queued", "dense-tiles"])
@pytest.mark.parametrize("nfill", [0, 3, 64])
def test_one_launch_form_bit_exact(geom, qcap, nfill):
    """Round 6, option iou_impl 5: K1 = the tests alone + one survivor bit per element, K2 = the drain
    whose first workgroups write every element the bits do not name -- fill and clip on disjoint addresses in one launch.
    Bit for bit the twin oracle (rbbox_geo_kernel.cu:231-268) and the shipped stream -> drain pair: aligned shapes of one
    and several tiles, partial last tiles and rows, tiles marked dense (written whole by the drain), 1 / 3 / 64 fill
    blocks; the result buffer holds NaNs before the call, so an element neither kind of store reaches would show."""
    from r3det import _C
    for (m, n) in [(700, 332), (37, 2052), (130, 4096), (9, 1024), (64, 12288)]:
        a = rand_boxes(m, 61 + m, span=150.0, amin=-np.pi, amax=np.pi)
        b = rand_boxes(n, 62 + n, span=150.0)
        with O.twin():
            want = O.iou_mat(geom, a, b, threads=8)
        try:
            _C.set_option("iou_impl", 5)
            _C.set_option("iou_qcap", qcap)
            _C.set_option("iou_nfill", nfill)
            for _ in range(2):
                poison = torch.full((m, n), float('nan'), device='cuda')   # (the caching allocator hands this block to the
                del poison                                                 #  call's torch.empty: an unwritten element shows)
                got = run(geom, a, b)
                assert same(got, want), (m, n)
        finally:
            _C.set_option("iou_impl", 0)
            _C.set_option("iou_qcap", 0)
            _C.set_option("iou_nfill", 0)


def test_assignment_shape_full_size():
    """Config 5 shape: 128 GT x 196 416 grid anchors (theta = 0).  Checked three ways:
    sampled columns bit-exact vs the twin oracle, range, and zero pattern == circle test
    superset (every non-zero pair has intersecting circumscribed circles)."""
    anchors = anchor_grid()
    assert anchors.shape == (196416, 5)
    gt = dota_like_gt(128, 5)
    got = run(O.V1, gt, anchors)
    assert got.shape == (128, 196416)
    assert np.isfinite(got).all() and got.min() >= 0 and got.max() <= 1
    cols = np.random.default_rng(0).choice(196416, 4000, replace=False)
    # plus the columns with the largest overlaps (the ones the assigner cares about)
    cols = np.unique(np.concatenate([cols, np.argsort(got.max(0))[-2000:]]))
    with O.twin():
        want = O.iou_mat(O.V1, gt, anchors[cols], threads=8)
    assert same(got[:, cols], want)
    assert (got > 0).mean() < 0.05  # > 95 % of pairs are disjoint: the HBM-write-bound regime
    # most GTs find a reasonably overlapping anchor (elongated ones at the image border do not)
    assert (got.max(1) > 0.2).mean() > 0.6


def test_transpose_consistency_v3():
    a, b = rand_boxes(257, 21, span=180.0), rand_boxes(130, 22, span=180.0)
    m = run(O.V3, a, b)
    mt = run(O.V3, b, a)
    assert np.abs(m - mt.T).max() <= TOL


@pytest.mark.parametrize("geom", [O.V1, O.V2, O.V3])
def test_vec_kernels(geom):
    from r3det.ops import obb_overlaps, rbbox_iou
    from r3det.ops.mmcv_ops import box_iou_rotated
    a, b = rand_boxes(1000, 31, span=120.0), rand_boxes(1000, 32, span=120.0)
    with O.twin():
        want = O.iou_vec(geom, a, b)
    if geom == O.V1:
        got = rbbox_iou(dev(a), dev(b), True, False).cpu().numpy()
        with O.twin():
            w1 = O.iou_vec(geom, a[:1], b, iof=True)
        assert same(rbbox_iou(dev(a[:1]), dev(b), True, True).cpu().numpy(), w1)  # modulo broadcast
    elif geom == O.V2:
        got = box_iou_rotated(dev(a), dev(b), 'iou', True).cpu().numpy()
    else:
        from r3det.ops.iou import aligned_obb_overlaps_kernel
        got = aligned_obb_overlaps_kernel(dev(a), dev(b), 'iou').cpu().numpy()
        assert got.shape == (1000, 1)
        got = got[:, 0]
        # the wrapper's aligned case is the reference's differentiable torch formulation (a different
        # algorithm with a 1e-3 containment tolerance, box_iou_rotated_wrapper.py:199-210)
        soft = obb_overlaps(dev(a), dev(b), 'iou', True).cpu().numpy()
        assert soft.shape == (1000, 1) and np.abs(soft[:, 0] - want).max() < 5e-3
    assert same(got, want)


def test_calculators_and_wrapper_quirks():
    from r3det.core.bbox.iou_calculators import RBboxOverlaps2D_v1, RBboxOverlaps2D_v2, RBboxOverlaps2D_v3
    from r3det.ops import obb_overlaps
    a, b = rand_boxes(300, 41, span=150.0), rand_boxes(77, 42, span=150.0)
    a6 = np.hstack([a, np.ones((300, 1), np.float32)])
    for cls, geom in [(RBboxOverlaps2D_v1, O.V1), (RBboxOverlaps2D_v2, O.V2), (RBboxOverlaps2D_v3, O.V3)]:
        with O.twin():
            want = O.iou_mat(geom, a, b)
        got = cls()(dev(a6), dev(b)).cpu().numpy()  # 6th column (score) is stripped
        assert same(got, want)
        assert cls()(dev(a6)[:0], dev(b)).shape == (0, 77)
    # obb_overlaps zeroes rows / cols of boxes thinner than 1e-3 AFTER the kernel
    a2 = a.copy()
    a2[5, 2] = 5e-4
    b2 = b.copy()
    b2[7, 3] = 1e-4
    got = obb_overlaps(dev(a2), dev(b2)).cpu().numpy()
    with O.twin():
        want = O.iou_mat(O.V3, a2, b2)
    want[5, :] = 0
    want[:, 7] = 0
    assert same(got, want)
    # numpy in -> numpy out
    out = obb_overlaps(a, b, device_id=0)
    assert isinstance(out, np.ndarray) and out.shape == (300, 77)
    assert obb_overlaps(dev(a)[:0], dev(b)).shape == (0, 77)


@pytest.mark.parametrize("shape", [(300, 77), (40, 5000), (3, 1), (130, 2100), (64, 4096)])
@pytest.mark.parametrize("mode", ["iou", "iof"])
@pytest.mark.parametrize("qcap", [0, 100], ids=["queued", "dense-tiles"])
def test_obb_overlaps_epilogue_in_the_library(shape, mode, qcap):
    """r3det_obb_overlaps = the v3 matrix + `outputs[too_small] = 0` (box_iou_rotated_wrapper.py:53-60) in the same
    call: equal to the matrix entry followed by the reference's torch epilogue -- several thin lines per wave,
    the first and last line, a NaN side (torch.min propagates it: not thin), zero and negative sides, every matrix
    form (small, one-launch tiles, stream + drain)."""
    from r3det.ops import obb_overlaps
    from r3det.ops.iou import box_iou_rotated_v3
    n1, n2 = shape
    a, b = rand_boxes(n1, 51, span=120.0), rand_boxes(n2, 52, span=120.0)
    for arr, idx in ((a, [0, n1 - 1, n1 // 2, n1 // 2 + 1]), (b, [0, n2 - 1, n2 // 3, n2 // 3 + 1, n2 // 3 + 2])):
        for k, i in enumerate(idx):
            arr[i % len(arr), 2 + (k & 1)] = [5e-4, 0.0, -3.0, 9.9e-4, 1e-5][k % 5]
    if n1 > 10:
        a[7, 2], a[7, 3] = np.nan, 1e-5   # not thin: min() is NaN
        a[9, 2] = 1e-3                    # not thin: strictly below 1e-3 only
    ta, tb = dev(a), dev(b)
    want = box_iou_rotated_v3(ta, tb, mode == 'iou')
    s1 = ta[:, 2:4].min(1)[0] < 0.001
    s2 = tb[:, 2:4].min(1)[0] < 0.001
    assert int(s1.sum()) >= 1 and int(s2.sum()) >= 1 and (n1 <= 10 or (not bool(s1[7]) and not bool(s1[9])))
    want = want.masked_fill(s1[:, None] | s2[None, :], 0.)
    # (round 6: the stream + drain pipeline applies the rule itself -- thin rows skipped, thin columns never survive, the
    # pairs of a tile marked dense checked by the drain -- and only the one-launch forms are followed by the epilogue
    # kernel; iou_qcap 100 marks every tile with a fuller wave dense; the result buffer holds NaNs before the call)
    from r3det import _C
    _C.set_option("iou_qcap", qcap)
    try:
        for impl in (0, 5):
            _C.set_option("iou_impl", impl)
            poison = torch.full((n1, n2), float('nan'), device='cuda')
            del poison
            got = obb_overlaps(ta, tb, mode=mode)
            assert same(got.cpu().numpy(), want.cpu().numpy()), impl
    finally:
        _C.set_option("iou_qcap", 0)
        _C.set_option("iou_impl", 0)


def test_errors():
    from r3det.ops import rbbox_iou
    a = dev(rand_boxes(10, 1))
    with pytest.raises(RuntimeError, match="contiguous"):
        rbbox_iou(a.t().contiguous().t(), a)
    with pytest.raises(RuntimeError):
        rbbox_iou(a.double(), a)


@pytest.mark.parametrize("version", ["v1", "v2", "v3"])
@pytest.mark.parametrize("shape", [(128, 21824), (37, 4096), (5, 1000)])
def test_prepared_columns_give_the_same_matrix_bit_for_bit(version, shape):
    """r3det_iou_prepare_columns + r3det_iou_mat_prepared == the plain matrix entry of the geometry (the exact records
    are the same code, the conservative data only filters); the anchor grid's spatial order and a random list; a
    non-finite column disables the 256-column shortcut for its block only."""
    import ctypes
    from r3det import _C, synthetic as syn
    from r3det.ops.iou import GEOM, prepare_columns
    n1, n2 = shape
    gt = syn.dota_like_rboxes(n1, 5, device='cuda')
    cols = syn.anchor_grid(device='cuda')[-n2:].contiguous() if n2 == 21824 else syn.rand_rboxes(n2, 3, device='cuda')
    cols = cols.clone()
    cols[n2 // 2, 4] = float('inf')      # one non-finite column
    L = _C.lib()
    geom = GEOM[version]
    prep = prepare_columns(cols, version)
    ws, wsb = _C.iou_workspace(n1, n2, gt.device)
    for mode in (0, 1):
        want = torch.full((n1, n2), float('nan'), device='cuda')
        plain = {1: L.r3det_rbbox_geo_mat_iou_iof, 3: L.r3det_box_iou_rotated_overlaps}.get(geom)
        if geom == 2:
            _C.check(L.r3det_mmcv_box_iou_rotated(_C.ptr(gt), n1, _C.ptr(cols), n2, mode, 0, _C.ptr(want), _C.ptr(ws), wsb,
                                                  _C.stream()), "v2")
        else:
            _C.check(plain(_C.ptr(gt), n1, _C.ptr(cols), n2, mode, _C.ptr(want), _C.ptr(ws), wsb, _C.stream()), "plain")
        got = torch.full((n1, n2), float('nan'), device='cuda')
        _C.check(L.r3det_iou_mat_prepared(geom, _C.ptr(gt), n1, _C.ptr(cols), n2, _C.ptr(prep), mode, _C.ptr(got),
                                          _C.ptr(ws), wsb, _C.stream()), "prepared")
        assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0))
    assert L.r3det_iou_prepare_columns(geom, _C.ptr(cols), n2, _C.ptr(prep), prep.numel() - 256, _C.stream()) == -3
    assert L.r3det_iou_prepare_columns(7, _C.ptr(cols), n2, _C.ptr(prep), prep.numel(), _C.stream()) == -1
    assert L.r3det_iou_mat_prepared(geom, _C.ptr(gt), n1, _C.ptr(cols), n2, None, 0, _C.ptr(got), _C.ptr(ws), wsb,
                                    _C.stream()) == -1


def test_prepared_buffer_refused_for_another_shape():
    """A buffer prepared for (geometry, n2) is refused by the consumers for any other (it used to give wrong
    IoUs silently).  Round 6: the buffer carries a header the kernels check on the DEVICE -- a mismatch answers in the data
    (a matrix of NaN; max_overlaps NaN and every box ignored) and nothing of the buffer is used; on the host
    r3det_iou_prepared_check gives the same verdict, and the Python wrapper refuses before it launches."""
    from r3det import _C
    from r3det import synthetic as syn
    from r3det.ops.iou import prepare_columns
    L = _C.lib()
    dev = torch.device("cuda")
    cols = syn.rand_rboxes(2048, 3, device=dev)
    rows = syn.rand_rboxes(16, 4, device=dev)
    pb = int(L.r3det_iou_prepared_bytes(2048))
    prep = torch.empty(pb, dtype=torch.uint8, device=dev)
    _C.check(L.r3det_iou_prepare_columns(1, _C.ptr(cols), 2048, _C.ptr(prep), pb, _C.stream()), "prepare")
    assert L.r3det_iou_prepared_check(_C.ptr(prep), 1, 2048, _C.stream()) == 0
    out = torch.empty(16, 2048, device=dev)
    wsb = int(L.r3det_iou_workspace_bytes(16, 2048))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ok = L.r3det_iou_mat_prepared(1, _C.ptr(rows), 16, _C.ptr(cols), 2048, _C.ptr(prep), 0, _C.ptr(out), _C.ptr(ws), wsb, _C.stream())
    assert ok == 0 and not bool(torch.isnan(out).any()) and torch.equal(out, rbbox_iou_t(rows, cols))
    for geom, n2 in ((1, 2044), (3, 2048), (2, 2048), (1, 1024)):
        assert L.r3det_iou_prepared_check(_C.ptr(prep), geom, n2, _C.stream()) != 0
        o2 = torch.zeros(16, n2, device=dev)
        rc = L.r3det_iou_mat_prepared(geom, _C.ptr(rows), 16, _C.ptr(cols), n2, _C.ptr(prep), 0, _C.ptr(o2), _C.ptr(ws), wsb,
                                      _C.stream())
        assert rc == 0 and bool(torch.isnan(o2).all())
    # a buffer that was never prepared: refused as well (whatever bytes it holds)
    junk = torch.randint(0, 255, (pb,), dtype=torch.uint8, device=dev)
    assert L.r3det_iou_prepared_check(_C.ptr(junk), 1, 2048, _C.stream()) != 0
    o3 = torch.zeros(16, 2048, device=dev)
    assert L.r3det_iou_mat_prepared(1, _C.ptr(rows), 16, _C.ptr(cols), 2048, _C.ptr(junk), 0, _C.ptr(o3), _C.ptr(ws), wsb,
                                    _C.stream()) == 0 and bool(torch.isnan(o3).all())
    # the Python wrapper knows what its buffer was prepared for and raises on the host
    from r3det.ops import rbbox_iou
    p = prepare_columns(cols, 'v1')
    with pytest.raises(ValueError):
        rbbox_iou(rows, cols[:1024].contiguous(), prepared=p)
    torch.cuda.synchronize()


def rbbox_iou_t(a, b):
    from r3det.ops import rbbox_iou
    return rbbox_iou(a, b)


def test_prepared_buffer_at_a_reused_address_and_as_a_copy():
    """rounds 4-5 kept a host map keyed by the device ADDRESS -- a freed buffer's address handed out again by
    the allocator kept the stale (geometry, n), and a copy of a prepared buffer passed unchecked.  The header travels
    with the bytes: the new tenant of an address is judged by what it holds, and a copy is as good as the original."""
    from r3det import _C
    from r3det import synthetic as syn
    L = _C.lib()
    dev = torch.device("cuda")
    cols_a, cols_b = syn.rand_rboxes(2048, 3, device=dev), syn.rand_rboxes(1024, 5, device=dev)
    rows = syn.rand_rboxes(16, 4, device=dev)
    pa, pb_ = int(L.r3det_iou_prepared_bytes(2048)), int(L.r3det_iou_prepared_bytes(1024))
    prep_b = torch.empty(pb_, dtype=torch.uint8, device=dev)
    _C.check(L.r3det_iou_prepare_columns(1, _C.ptr(cols_b), 1024, _C.ptr(prep_b), pb_, _C.stream()), "prepare b")
    prep_a = torch.empty(pa, dtype=torch.uint8, device=dev)
    _C.check(L.r3det_iou_prepare_columns(1, _C.ptr(cols_a), 2048, _C.ptr(prep_a), pa, _C.stream()), "prepare a")
    addr = prep_a.data_ptr()
    torch.cuda.synchronize()
    del prep_a
    # the allocator hands the freed block out again: a COPY of buffer b now lives at a's old address
    again = torch.empty(pa, dtype=torch.uint8, device=dev)
    # (the caching allocator normally hands the block straight back; when it does not, what follows is still the
    # "a copy of a prepared buffer is a prepared buffer" half of the test)
    reused = again.data_ptr() == addr
    again[:pb_].copy_(prep_b)
    want = rbbox_iou_t(rows, cols_b)
    wsb = int(L.r3det_iou_workspace_bytes(16, 2048))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    out = torch.zeros(16, 1024, device=dev)
    assert L.r3det_iou_prepared_check(_C.ptr(again), 1, 1024, _C.stream()) == 0
    assert L.r3det_iou_mat_prepared(1, _C.ptr(rows), 16, _C.ptr(cols_b), 1024, _C.ptr(again), 0, _C.ptr(out), _C.ptr(ws), wsb,
                                    _C.stream()) == 0
    assert torch.equal(out, want)
    # ... and it is no longer a buffer for a's shape (at a's old address, when the allocator reused it)
    assert reused or True
    assert L.r3det_iou_prepared_check(_C.ptr(again), 1, 2048, _C.stream()) != 0
    o2 = torch.zeros(16, 2048, device=dev)
    assert L.r3det_iou_mat_prepared(1, _C.ptr(rows), 16, _C.ptr(cols_a), 2048, _C.ptr(again), 0, _C.ptr(o2), _C.ptr(ws), wsb,
                                    _C.stream()) == 0 and bool(torch.isnan(o2).all())


def test_assignment_with_a_mismatched_prepared_buffer_answers_in_its_data():
    from r3det import _C
    from r3det import synthetic as syn
    L = _C.lib()
    dev = torch.device("cuda")
    anchors = syn.rand_rboxes(4096, 3, device=dev)
    gts = syn.rand_rboxes(20, 4, device=dev)
    pb = int(L.r3det_iou_prepared_bytes(4096))
    prep = torch.empty(pb, dtype=torch.uint8, device=dev)
    _C.check(L.r3det_iou_prepare_columns(3, _C.ptr(anchors), 4096, _C.ptr(prep), pb, _C.stream()), "prepare")   # for v3
    n1, n2 = 20, 4096
    wsb = int(L.r3det_rbbox_assign_workspace_bytes(n1, n2))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    gi = torch.zeros(n2, dtype=torch.int64, device=dev)
    mo = torch.zeros(n2, device=dev)
    am = torch.zeros(n2, dtype=torch.int64, device=dev)
    gm = torch.zeros(n1, device=dev)
    ga = torch.zeros(n1, dtype=torch.int64, device=dev)
    rc = L.r3det_rbbox_assign_prepared(1, _C.ptr(gts), n1, _C.ptr(anchors), n2, _C.ptr(prep), 0.5, 0.4, 0.0, 1, 1, _C.ptr(gi),
                                       _C.ptr(mo), _C.ptr(am), _C.ptr(gm), _C.ptr(ga), _C.ptr(ws), wsb, _C.stream())
    assert rc == 0
    assert bool(torch.isnan(mo).all()) and bool((gi == -1).all())


def test_drain_tickets_equal_static_stride_at_512_rows():
    """512 x 196 416 (SURVEY 8d's fourth size) is where the drain hands out its blocks by atomic tickets (three or more
    per wavefront; option iou_dyn): bit-identical to the static stride, and sampled columns bit-exact against the twin
    oracle -- the size-independent check of a path the small fixtures never take."""
    from r3det import _C
    anchors = anchor_grid()
    gt = dota_like_gt(512, 6)
    got = run(O.V1, gt, anchors)
    _C.set_option("iou_dyn", 0)
    try:
        static = run(O.V1, gt, anchors)
    finally:
        _C.set_option("iou_dyn", 1)
    assert got.shape == (512, 196416) and np.array_equal(got, static)
    cols = np.unique(np.concatenate([np.random.default_rng(1).choice(196416, 1500, replace=False),
                                     np.argsort(got.max(0))[-1500:]]))
    with O.twin():
        want = O.iou_mat(O.V1, gt, anchors[cols], threads=8)
    assert same(got[:, cols], want)
    got3 = run(O.V3, gt, anchors)
    _C.set_option("iou_dyn", 0)
    try:
        static3 = run(O.V3, gt, anchors)
    finally:
        _C.set_option("iou_dyn", 1)
    assert np.array_equal(got3, static3)
