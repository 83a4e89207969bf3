"""GPU parity of round 6's tap tables: the backward's index as a by-product of the forward launch.

The reference's backward scatters with atomics and needs no index (fr/src/feature_refine_kernel.cu:165-230); this
library's deterministic gather does.  A training step's forward launches write, per level, the clamped sample point of
every position (the clamps of bilinear_interpolate, feature_refine_kernel.cu:22-47) and the index kernel scans that
table instead of the 20-byte box records.  Checked here:
  * the table the channels_last launches write == r3det_feature_refine_prepare's table of the same boxes (bit for bit)
    == the oracle-side restatement of the clamps;
  * the launches' outputs do not change (bit for bit);
  * the index built from tables == the index built from the boxes, BYTE for byte, both layouts, incl. boxes outside the
    map, NaN coordinates and piles on one cell;
  * the autograd functions give bit-identical gradients with the tables on and off.
"""
import numpy as np
import pytest
import torch

from helpers import fr_boxes
from test_gpu_fr import dev

pytestmark = pytest.mark.gpu
CL = torch.channels_last
PYRAMID = [(128, 8), (64, 16), (32, 32), (16, 64), (8, 128)]


def clamp_table(boxes, N, H, W, scale):
    """cell_tap in numpy fp32: [N][y: HW][x: HW] (row <- x_ctr, column <- y_ctr, feature_refine_kernel.cu:125-131)."""
    b = boxes.reshape(N, H * W, 5)
    y = (b[..., 0] * np.float32(scale)).astype(np.float32)
    x = (b[..., 1] * np.float32(scale)).astype(np.float32)
    with np.errstate(invalid="ignore"):
        bad = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
        y = np.where(y <= 0, np.float32(0), y)
        x = np.where(x <= 0, np.float32(0), x)
        y = np.where(np.nan_to_num(y, nan=0.0).astype(np.int64) >= H - 1, np.float32(H - 1), y)
        x = np.where(np.nan_to_num(x, nan=0.0).astype(np.int64) >= W - 1, np.float32(W - 1), x)
    y = np.where(bad, np.float32(H + 1), y).astype(np.float32)
    x = np.where(bad, np.float32(0), x).astype(np.float32)
    return np.concatenate([y, x], 1).reshape(-1)


def level_inputs(N, C, levels, seed, field="regular"):
    r = np.random.default_rng(seed)
    feats, boxes = [], []
    for hw, stride in levels:
        feats.append(dev(r.normal(size=(N, C, hw, hw)).astype(np.float32)).contiguous(memory_format=CL))
        b = fr_boxes(N, hw, hw, stride, seed + hw, adversarial=(field == "adversarial"))
        if field == "piles":        # every 4 x 4 block of positions regresses to one centre
            g = np.floor(b[:, :2] / (4 * stride)) * 4 * stride + 2 * stride
            b[:, :2] = g + r.normal(0, 0.3 * stride, g.shape)
        if field == "nan":
            b[::97, 0] = np.nan
            b[5::131, 1] = np.nan
        boxes.append(dev(b.astype(np.float32)))
    return feats, boxes


@pytest.mark.parametrize("field", ["regular", "adversarial", "piles", "nan"])
@pytest.mark.parametrize("N,C,levels", [(2, 16, PYRAMID), (4, 8, PYRAMID[:2]), (1, 8, [(20, 16), (8, 64)])])
def test_forward_levels_nhwc_writes_the_tap_tables(field, N, C, levels):
    from r3det.ops.feature_refine import fr_forward_levels_nhwc, fr_prepare, tap_tables
    feats, boxes = level_inputs(N, C, levels, 3, field)
    scales = [1 / s for _, s in levels]
    shapes = [(hw, hw) for hw, _ in levels]
    plain = [torch.empty_like(f) for f in feats]
    assert fr_forward_levels_nhwc(feats, boxes, scales, 1, plain)
    tabs = tap_tables(N, shapes, feats[0].device)
    for t in tabs:
        t.fill_(float("nan"))
    outs = [torch.empty_like(f) for f in feats]
    assert fr_forward_levels_nhwc(feats, boxes, scales, 1, outs, tabs)
    for o, p in zip(outs, plain):
        assert torch.equal(o, p) or (field == "nan" and torch.equal(torch.nan_to_num(o), torch.nan_to_num(p)))
    for (hw, stride), b, t in zip(levels, boxes, tabs):
        want = clamp_table(b.cpu().numpy(), N, hw, hw, 1 / stride)
        got = t.cpu().numpy()
        assert got.shape == want.shape
        assert np.array_equal(got, want, equal_nan=True)
        prep = fr_prepare(b, N, hw, hw, 1 / stride, 1)       # (128 / 64 maps: the NCHW path's table kernel)
        if prep is not None:
            assert np.array_equal(prep.cpu().numpy(), got, equal_nan=True)


@pytest.mark.parametrize("field", ["regular", "adversarial", "piles", "nan"])
@pytest.mark.parametrize("N,levels", [(4, PYRAMID), (2, PYRAMID[:1]), (1, [(20, 16), (8, 64)])])
def test_index_from_tables_is_byte_identical_to_the_index_from_boxes(field, N, levels):
    from r3det.ops.feature_refine import (fr_backward_index_levels, fr_backward_nhwc_index_levels,
                                          fr_forward_levels_nhwc, tap_tables)
    C = 8
    feats, boxes = level_inputs(N, C, levels, 5, field)
    scales = [1 / s for _, s in levels]
    shapes = [(hw, hw) for hw, _ in levels]
    tabs = tap_tables(N, shapes, feats[0].device)
    assert fr_forward_levels_nhwc(feats, boxes, scales, 1, [torch.empty_like(f) for f in feats], tabs)
    # (every fill value: bytes neither form writes must not matter; a second box-form build first -- a band beyond the
    # sorted form's capacity takes the general form, whose long lists keep their LDS-atomic arrival order: such a field is
    # not byte-stable run to run in EITHER form, and is compared through the gradients in the tests below instead)
    def same(nhwc, Cg, nbytes):
        for fill in (0, 255):
            wa = torch.full((max(nbytes, 16),), fill, dtype=torch.uint8, device="cuda")
            wb, wc = wa.clone(), wa.clone()
            _index_into(wa, boxes, None, N, shapes, scales, nhwc=nhwc, C=Cg)
            _index_into(wc, boxes, None, N, shapes, scales, nhwc=nhwc, C=Cg)
            _index_into(wb, boxes, tabs, N, shapes, scales, nhwc=nhwc, C=Cg)
            if not torch.equal(wa, wc):
                assert field == "piles", "the box form itself is not byte-stable on this field"
                continue
            if not torch.equal(wa, wb):
                d = (wa != wb).nonzero().flatten()
                raise AssertionError(f"{'nhwc' if nhwc else 'nchw'} C={Cg} fill={fill}: {d.numel()} bytes differ, first at "
                                     f"{int(d[0])} of {nbytes}, last at {int(d[-1])}")
    # channels_last gathers' CSR index
    a = fr_backward_nhwc_index_levels(boxes, N, shapes, scales, 1, None)
    b = fr_backward_nhwc_index_levels(boxes, N, shapes, scales, 1, tabs)
    assert a is not None and b is not None and a[1] == b[1]
    same(True, 8, a[1])
    # NCHW gathers' CSR + SELL index (C decides the interleave)
    for Cg in (8, 256):
        same(False, Cg, fr_backward_index_levels(boxes, N, Cg, shapes, scales, 1, None)[1])


def _index_into(ws, boxes, tabs, N, shapes, scales, nhwc, C=8):
    import ctypes

    from r3det import _C
    from r3det.ops.feature_refine import _ptr_array
    n = len(boxes)
    arr_i = ctypes.c_int * n
    H, W = arr_i(*[h for h, _ in shapes]), arr_i(*[w for _, w in shapes])
    sc = (ctypes.c_float * n)(*[float(s) for s in scales])
    L = _C.lib()
    if nhwc:
        if tabs is None:
            rc = L.r3det_feature_refine_backward_nhwc_index_levels(n, _ptr_array(boxes), N, H, W, sc, 1, _C.ptr(ws),
                                                                   ws.numel(), _C.stream())
        else:
            rc = L.r3det_feature_refine_backward_nhwc_index_levels_tab(n, _ptr_array(boxes), _ptr_array(tabs), N, H, W, sc,
                                                                       1, _C.ptr(ws), ws.numel(), _C.stream())
    else:
        if tabs is None:
            rc = L.r3det_feature_refine_backward_index_levels(n, _ptr_array(boxes), N, C, H, W, sc, 1, _C.ptr(ws),
                                                              ws.numel(), _C.stream())
        else:
            rc = L.r3det_feature_refine_backward_index_levels_tab(n, _ptr_array(boxes), _ptr_array(tabs), N, C, H, W, sc, 1,
                                                                  _C.ptr(ws), ws.numel(), _C.stream())
    _C.check(rc, "index")
    torch.cuda.synchronize()


def test_a_level_without_a_table_reads_its_boxes():
    """tables[l] = NULL: that level from its boxes (the NCHW path has tables for its 128 / 64 levels only)."""
    from r3det.ops.feature_refine import fr_backward_index_levels, fr_prepare
    N, C = 2, 256
    _, boxes = level_inputs(N, 8, PYRAMID, 9)
    scales = [1 / s for _, s in PYRAMID]
    shapes = [(hw, hw) for hw, _ in PYRAMID]
    tabs = [fr_prepare(b, N, hw, hw, 1 / s, 1) for b, (hw, s) in zip(boxes, PYRAMID)]
    assert tabs[0] is not None and tabs[1] is not None and all(t is None for t in tabs[2:])
    a = fr_backward_index_levels(boxes, N, C, shapes, scales, 1, None)
    wa = torch.zeros(a[1], dtype=torch.uint8, device="cuda")
    wb = wa.clone()
    _index_into(wa, boxes, None, N, shapes, scales, nhwc=False, C=C)
    _index_into(wb, boxes, tabs, N, shapes, scales, nhwc=False, C=C)
    assert torch.equal(wa, wb)


def test_tab_entry_points_refuse_what_they_cannot_do():
    import ctypes

    from r3det import _C
    from r3det.ops.feature_refine import _ptr_array, tap_tables
    N, C = 1, 8
    feats, boxes = level_inputs(N, C, PYRAMID[:1], 2)
    L = _C.lib()
    H = W = (ctypes.c_int * 1)(128)
    sc = (ctypes.c_float * 1)(0.125)
    out = [torch.empty_like(feats[0])]
    tabs = tap_tables(N, [(128, 128)], "cuda")
    assert int(L.r3det_fr_tap_table_bytes(N, 128, 128)) == N * 128 * 128 * 8 and int(L.r3det_fr_tap_table_bytes(0, 1, 1)) == 0
    # points = 5 has five sample points per position: no table form
    assert L.r3det_feature_refine_forward_levels_nhwc_tab(1, _ptr_array(feats), _ptr_array(boxes), N, C, H, W, sc, 5,
                                                          _ptr_array(out), _ptr_array(tabs), _C.stream()) == -1
    assert L.r3det_feature_refine_forward_levels_nhwc_tab(1, _ptr_array(feats), _ptr_array(boxes), N, C, H, W, sc, 1,
                                                          _ptr_array(out), None, _C.stream()) == -1
    ws = torch.empty(int(L.r3det_fr_backward_nhwc_levels_workspace_bytes(1, N, H, W, 1)), dtype=torch.uint8, device="cuda")
    assert L.r3det_feature_refine_backward_nhwc_index_levels_tab(1, _ptr_array(boxes), None, N, H, W, sc, 1, _C.ptr(ws),
                                                                 ws.numel(), _C.stream()) == -1
    assert L.r3det_feature_refine_backward_nhwc_index_levels_tab(1, _ptr_array(boxes), _ptr_array(tabs), N, H, W, sc, 5,
                                                                 _C.ptr(ws), ws.numel(), _C.stream()) == -1
    # a misaligned table
    off = [tabs[0].new_empty(tabs[0].numel() + 1)[1:]]
    assert L.r3det_feature_refine_backward_nhwc_index_levels_tab(1, _ptr_array(boxes), _ptr_array(off), N, H, W, sc, 1,
                                                                 _C.ptr(ws), ws.numel(), _C.stream()) == -1


@pytest.mark.parametrize("layout", ["channels_last", "nchw"])
@pytest.mark.parametrize("fn", ["module_levels", "levels", "single"])
def test_autograd_gradients_are_bit_identical_with_and_without_the_tables(layout, fn):
    import r3det.ops.feature_refine as FRM
    N, C = 2, 16
    levels = PYRAMID if fn != "single" else PYRAMID[:1]
    feats, boxes = level_inputs(N, C, levels, 21, "piles")
    if layout == "nchw":
        feats = [f.contiguous() for f in feats]
    scales = [1 / s for _, s in levels]
    r = torch.Generator(device="cuda").manual_seed(3)
    ups = [torch.randn(f.shape, device="cuda", generator=r).contiguous(memory_format=CL if layout != "nchw" else
                                                                      torch.contiguous_format) for f in feats]

    def run(on):
        FRM.TRAIN_TAP_TABLES = on
        try:
            xs = [f.clone(memory_format=torch.preserve_format).requires_grad_(True) for f in feats]
            if fn == "module_levels":
                aa = [f.clone(memory_format=torch.preserve_format).requires_grad_(True) for f in feats]
                outs = FRM.feature_refine_module_levels(aa, [x * 0.5 for x in xs], xs, boxes, scales, 1)
            elif fn == "levels":
                outs = FRM.feature_refine_levels(xs, boxes, scales, 1)
            else:
                table = FRM.fr_prepare(boxes[0], N, 128, 128, scales[0], 1) if layout == "nchw" else None
                outs = [FRM.feature_refine(xs[0], boxes[0], scales[0], 1, table)]
            torch.autograd.backward(outs, ups[:len(outs)])
            return [o.detach().clone() for o in outs], [x.grad.clone() for x in xs]
        finally:
            FRM.TRAIN_TAP_TABLES = True

    o1, g1 = run(True)
    o0, g0 = run(False)
    for a, b in zip(o1 + g1, o0 + g0):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------------------
# r3det_feature_refine_module_levels: the NCHW module tail of all levels in one call (round 6)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,C", [(2, 256), (4, 128), (1, 512), (4, 16), (1, 6)])
@pytest.mark.parametrize("field", ["regular", "adversarial"])
def test_module_levels_nchw_equals_the_three_step_form_bit_for_bit(N, C, field):
    """out_l = x_l + fr(a_l + b_l) (feature_refine_module.py:121-126) from ONE library call against the reference's own
    sequence of operations: torch add, the sampler (r3det_feature_refine_forward, itself pinned to the oracle in
    test_gpu_fr.py), torch add."""
    from r3det.ops.feature_refine import fr_forward, fr_module_levels
    feats, boxes = level_inputs(N, C, PYRAMID, 17, field)
    xs = [f.contiguous() for f in feats]
    g = torch.Generator(device="cuda").manual_seed(5)
    a = [torch.randn(x.shape, device="cuda", generator=g) for x in xs]
    b = [torch.randn(x.shape, device="cuda", generator=g) for x in xs]
    scales = [1 / s for _, s in PYRAMID]
    outs = [torch.full_like(x, float("nan")) for x in xs]
    ws = fr_module_levels(a, b, xs, boxes, scales, 1, outs)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if C % 2 or N * C // 2 < cus:   # (the fused cell kernel wants two channels per workgroup on every compute unit: else
        assert ws is None           #  the call answers "not taken" before launching anything and the caller goes level by level)
        assert all(bool(torch.isnan(o).all()) for o in outs)
        return
    assert ws is not None
    for x, p, q, bx, s, o in zip(xs, a, b, boxes, scales, outs):
        m = p + q
        want = torch.empty_like(m)
        fr_forward(m, bx, s, 1, want)
        want += x
        assert torch.equal(o, want)


def test_module_levels_nchw_refuses_shapes_it_does_not_take_without_launching():
    from r3det.ops.feature_refine import fr_module_levels
    N, C = 1, 8
    levels = [(20, 16), (10, 32)]          # W % 4 != 0 at the second level
    feats, boxes = level_inputs(N, C, levels, 3)
    xs = [f.contiguous() for f in feats]
    outs = [torch.full_like(x, 7.0) for x in xs]
    assert fr_module_levels(xs, xs, xs, boxes, [1 / 16, 1 / 32], 1, outs) is None
    assert all(bool((o == 7.0).all()) for o in outs)
    assert fr_module_levels(xs, xs, xs, boxes, [1 / 16, 1 / 32], 5, outs) is None      # points = 5: no fused form


def test_module_forward_nchw_inference_takes_the_levels_call():
    """FeatureRefineModule.forward on NCHW maps without gradients == the per-level reference sequence."""
    from r3det.ops import FeatureRefineModule
    from r3det.ops.feature_refine import fr_forward
    N, C = 2, 256
    feats, boxes = level_inputs(N, C, PYRAMID, 23)
    xs = [f.contiguous() for f in feats]
    m = FeatureRefineModule(C, [s for _, s in PYRAMID]).cuda().eval()
    per_img = [[b.view(N, -1, 5)[i] for b in boxes] for i in range(N)]
    with torch.no_grad():
        got = m(xs, per_img)
        for x, bx, (hw, s), o in zip(xs, boxes, PYRAMID, got):
            mixed = m.conv_5_1(m.conv_1_5(x)) + m.conv_1_1(x)
            want = torch.empty_like(mixed)
            fr_forward(mixed.contiguous(), bx, 1 / s, 1, want)
            # (the convolutions are run twice here and MIOpen's fp32 kernels are not bit-stable run to run; the call's own
            # bit-exactness is the test above)
            assert torch.allclose(o, x + want, rtol=0, atol=2e-5)
