"""GPU parity: Feature Refinement forward / backward through the C ABI vs the oracle.
Forward: bit-exact vs the twin oracle (tolerance bar of north_star is 1e-5).  Backward: the library sums a cell's
contributions in its own fixed order (gather over the inverse tap index), the oracle in the reference's loop order
=> tolerance 1e-5 relative to the gradient scale, written below; run to run the library's results are bit-identical."""
import numpy as np
import pytest
import torch

from helpers import fr_boxes
from oracle import api as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(params=[1, 2, 10, 0], ids=["generic", "lds-plane", "cell", "auto"])
def impl(request):
    from r3det import _C
    _C.set_option("fr_impl", request.param)
    yield request.param
    _C.set_option("fr_impl", 0)


SHAPES = [(2, 8, 16, 16, 8), (1, 5, 7, 13, 16), (2, 3, 128, 128, 8), (1, 40, 32, 32, 32), (3, 2, 1, 1, 128),
          (1, 2, 8, 8, 128), (2, 5, 64, 64, 16), (1, 3, 32, 128, 8), (1, 2, 256, 64, 8), (2, 512, 64, 64, 16),
          (1, 512, 128, 128, 8)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("points", [1, 5])
@pytest.mark.parametrize("adversarial", [False, True])
def test_forward_bit_exact(impl, shape, points, adversarial):
    from r3det.ops.feature_refine import fr_forward
    N, C, H, W, stride = shape
    r = np.random.default_rng(5)
    feat = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 6, adversarial=adversarial)
    with O.twin():
        want = O.fr_forward(feat, boxes, 1 / stride, points, threads=8)
    out = torch.full((N, C, H, W), float('nan'), device='cuda')
    assert fr_forward(dev(feat), dev(boxes), 1 / stride, points, out) == 1
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("shape", SHAPES[:9])
@pytest.mark.parametrize("points", [1, 5])
def test_backward(impl, shape, points):
    from r3det.ops.feature_refine import fr_backward
    N, C, H, W, stride = shape
    r = np.random.default_rng(7)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 8)
    with O.twin():
        want = O.fr_backward(top, boxes, 1 / stride, points)
    tol = 1e-5 * max(1.0, np.abs(want).max())
    # reference calling convention: accumulate into a zero-filled buffer
    g = torch.zeros((N, C, H, W), device='cuda')
    fr_backward(dev(top), dev(boxes), 1 / stride, points, g)
    assert np.abs(g.cpu().numpy() - want).max() <= tol
    # accumulation really accumulates
    fr_backward(dev(top), dev(boxes), 1 / stride, points, g)
    assert np.abs(g.cpu().numpy() - 2 * want).max() <= 2 * tol
    # overwrite mode needs no zero fill
    g2 = torch.full((N, C, H, W), float('nan'), device='cuda')
    fr_backward(dev(top), dev(boxes), 1 / stride, points, g2, overwrite=True)
    assert np.abs(g2.cpu().numpy() - want).max() <= tol


@pytest.mark.parametrize("shape", [(1, 1024, 128, 128, 8), (1, 2048, 64, 64, 16), (3, 512, 64, 64, 16)])
def test_cell_kernel_long_channel_runs(shape):
    """The cell kernel's steady-state loop (G = 4 / 8 channels per workgroup: both prologue and both
    epilogue phases plus the rolling load pipeline in between), in automatic mode."""
    from r3det.ops.feature_refine import fr_forward
    N, C, H, W, stride = shape
    r = np.random.default_rng(11)
    feat = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 12, adversarial=True)
    with O.twin():
        want = O.fr_forward(feat, boxes, 1 / stride, 1, threads=8)
    for impl in (10, 0):
        from r3det import _C
        _C.set_option("fr_impl", impl)
        out = torch.full((N, C, H, W), float('nan'), device='cuda')
        fr_forward(dev(feat), dev(boxes), 1 / stride, 1, out)
        _C.set_option("fr_impl", 0)
        assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("shape", [(1, 4, 100, 100, 8), (2, 6, 100, 128, 8), (1, 2, 90, 100, 8)])
@pytest.mark.parametrize("adversarial", [False, True])
def test_backward_gather_planes_that_leave_wavefronts_without_slices(shape, adversarial):
    """Planes of 8 193 ... 16 383 cells whose two-channel form takes 16 slices per wavefront (the form with three index
    batches in flight and its first loads in front of the barrier): the last wavefronts own a partial set of slices or
    none at all and must still meet the barrier; against the oracle, overwrite and accumulate."""
    from r3det.ops.feature_refine import fr_backward
    N, C, H, W, stride = shape
    r = np.random.default_rng(23)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 9, adversarial=adversarial)
    with O.twin():
        want = O.fr_backward(top, boxes, 1 / stride, 1)
    tol = 1e-5 * max(1.0, np.abs(want).max())
    g = torch.full((N, C, H, W), float('nan'), device='cuda')
    fr_backward(dev(top), dev(boxes), 1 / stride, 1, g, overwrite=True)
    assert np.abs(g.cpu().numpy() - want).max() <= tol
    pre = r.normal(size=(N, C, H, W)).astype(np.float32)
    g2 = dev(pre)
    fr_backward(dev(top), dev(boxes), 1 / stride, 1, g2, overwrite=False)
    assert np.abs(g2.cpu().numpy() - (pre + want)).max() <= 2 * tol


@pytest.mark.parametrize("shape", [(1, 512, 128, 128, 8), (1, 1024, 128, 128, 8), (2, 512, 64, 64, 16),
                                   (3, 512, 64, 64, 16)])
@pytest.mark.parametrize("adversarial", [False, True])
def test_backward_gather_full_channel_counts(shape, adversarial):
    """The NCHW gather at the model's channel counts (two / four interleaved channels per staged cell, one to four
    channel groups per workgroup: the register-prefetched steady state) against the oracle and against the
    LDS-atomic plane kernel; twice: bit-identical run to run."""
    from r3det import _C
    from r3det.ops.feature_refine import fr_backward
    N, C, H, W, stride = shape
    r = np.random.default_rng(17)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 8, adversarial=adversarial)
    with O.twin():
        want = O.fr_backward(top[:, :8], boxes, 1 / stride, 1)       # the oracle is serial: first 8 channels
    tol = 1e-5 * max(1.0, np.abs(want).max())
    g = torch.full((N, C, H, W), float('nan'), device='cuda')
    fr_backward(dev(top), dev(boxes), 1 / stride, 1, g, overwrite=True)
    assert np.abs(g[:, :8].cpu().numpy() - want).max() <= tol
    again = torch.full((N, C, H, W), float('nan'), device='cuda')
    fr_backward(dev(top), dev(boxes), 1 / stride, 1, again, overwrite=True)
    assert torch.equal(g, again)
    # the index kernel re-lays the lists as SELL rows itself at these shapes; option frb_impl 3: the separate launch
    # from the CSR lists instead -- the same rows, bit for bit; 1: the general index form (another list order)
    for impl, exact in ((3, True), (5, True), (1, False)):
        _C.set_option("frb_impl", impl)
        other = torch.full((N, C, H, W), float('nan'), device='cuda')
        fr_backward(dev(top), dev(boxes), 1 / stride, 1, other, overwrite=True)
        _C.set_option("frb_impl", 0)
        assert torch.equal(g, other) if exact else (g - other).abs().max().item() <= tol
    _C.set_option("fr_impl", 2)
    g2 = torch.empty_like(g)
    fr_backward(dev(top), dev(boxes), 1 / stride, 1, g2, overwrite=True)
    _C.set_option("fr_impl", 0)
    assert (g - g2).abs().max().item() <= 1e-5 * max(1.0, g2.abs().max().item())


@pytest.mark.parametrize("shape", [(2, 512, 128, 128, 8), (2, 512, 64, 64, 16)])
@pytest.mark.parametrize("pile", [2, 9, 10, 16, 17, 300, -1])
def test_backward_piles_of_sources_on_one_cell(shape, pile):
    """`pile` positions of image 1 sample one cell (-1: every position of it), image 0 keeps its regular field.
    Lists up to the SELL capacity (32 entries) are summed from the slice layout, longer ones finish from the CSR
    array lane by lane: both in one call, against the oracle."""
    from r3det.ops.feature_refine import fr_backward
    N, C, H, W, stride = shape
    r = np.random.default_rng(pile + 100)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 21).reshape(N, H * W, 5)
    idx = np.arange(H * W) if pile < 0 else r.choice(H * W, pile, replace=False)
    boxes[1, idx, 0] = 37.0 * stride  # same sampled cell (row 37, column 11), different fractions
    boxes[1, idx, 1] = 11.6 * stride
    boxes[1, idx, 0] += r.uniform(0, 0.9 * stride, idx.size).astype(np.float32)
    boxes = boxes.reshape(-1, 5)
    with O.twin():
        want = O.fr_backward(top[:, :4], boxes, 1 / stride, 1)
    tol = 1e-5 * max(1.0, np.abs(want).max())
    g = torch.full((N, C, H, W), float('nan'), device='cuda')
    fr_backward(dev(top), dev(boxes), 1 / stride, 1, g, overwrite=True)
    assert np.abs(g[:, :4].cpu().numpy() - want).max() <= tol
    assert bool(torch.isfinite(g).all())


@pytest.mark.parametrize("shape", [(1, 512, 128, 128, 8), (2, 512, 64, 64, 16), (2, 8, 64, 64, 16), (1, 4, 32, 32, 32)])
def test_split_form_prepare_then_sample(shape):
    """r3det_feature_refine_prepare + _forward_prepared == the one-call form (and the autograd function
    falls back to the one-call form where the split form does not apply)."""
    from r3det.ops.feature_refine import feature_refine, fr_forward, fr_forward_prepared, fr_prepare
    N, C, H, W, stride = shape
    r = np.random.default_rng(21)
    feat = dev(r.normal(size=(N, C, H, W)).astype(np.float32))
    boxes = dev(fr_boxes(N, H, W, stride, 3, adversarial=True))
    want = torch.empty_like(feat)
    fr_forward(feat, boxes, 1 / stride, 1, want)
    table = fr_prepare(boxes, N, H, W, 1 / stride)
    assert (table is not None) == (H in (64, 128))
    if table is not None:
        out = torch.full_like(feat, float('nan'))
        if fr_forward_prepared(feat, table, out):
            assert C >= 256 and torch.equal(out, want)
        else:
            assert C < 256          # too few planes for the channel-group kernel
    assert torch.equal(feature_refine(feat, boxes, 1 / stride, 1, table), want)


def test_plane_too_large_falls_back_to_generic():
    """200 x 200 planes (160 KB padded) exceed the LDS budget: auto mode must still be right."""
    from r3det.ops.feature_refine import fr_forward
    N, C, H, W, stride = 1, 2, 200, 200, 8
    feat = np.random.default_rng(1).normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 2)
    with O.twin():
        want = O.fr_forward(feat, boxes, 1 / stride, 1, threads=8)
    out = torch.empty((N, C, H, W), device='cuda')
    fr_forward(dev(feat), dev(boxes), 1 / stride, 1, out)
    assert np.array_equal(out.cpu().numpy(), want)


def test_autograd_and_module_config3_shapes():
    """R3Det FRM on the 5 pyramid levels of a 1024 x 1024 input, N = 2 (reduced C to keep the
    CPU oracle in seconds): module output == x + FR(conv mix), gradients flow to x and convs."""
    from r3det.ops import FeatureRefineModule
    strides = [8, 16, 32, 64, 128]
    N, C = 2, 8
    torch.manual_seed(0)
    m = FeatureRefineModule(C, strides).cuda()
    m.init_weights()
    xs = [torch.randn(N, C, 1024 // s, 1024 // s, device='cuda', requires_grad=True) for s in strides]
    rois = [[dev(fr_boxes(1, 1024 // s, 1024 // s, s, 10 * i + j)) for j, s in enumerate(strides)]
            for i in range(N)]
    outs = m(xs, rois)
    for lvl, (x, o, s) in enumerate(zip(xs, outs, strides)):
        mixed = m.conv_5_1(m.conv_1_5(x)) + m.conv_1_1(x)
        boxes = torch.cat([rois[i][lvl] for i in range(N)]).cpu().numpy()
        with O.twin():
            want = O.fr_forward(mixed.detach().cpu().numpy(), boxes, 1 / s, 1, threads=8)
        assert np.abs((o - x).detach().cpu().numpy() - want).max() <= 1e-5
    sum(o.square().sum() for o in outs).backward()
    assert all(x.grad is not None and torch.isfinite(x.grad).all() for x in xs)
    assert m.conv_1_1.weight.grad.abs().sum() > 0


def test_gradcheck_against_oracle_backward():
    from r3det.ops.feature_refine import feature_refine
    N, C, H, W, stride = 1, 2, 6, 6, 8
    r = np.random.default_rng(3)
    x = dev(r.normal(size=(N, C, H, W)).astype(np.float32)).requires_grad_(True)
    boxes = fr_boxes(N, H, W, stride, 4)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    y = feature_refine(x, dev(boxes), 1 / stride, 5)
    y.backward(dev(top))
    want = O.fr_backward(top, boxes, 1 / stride, 5)
    assert np.abs(x.grad.cpu().numpy() - want).max() <= 1e-5 * max(1, np.abs(want).max())
    with pytest.raises(AssertionError):
        feature_refine(x, dev(boxes), 1 / stride, 3)  # points must be 1 or 5


def test_backward_ws_too_small_or_absent_uses_the_scatter_kernel():
    """r3det_feature_refine_backward_ws with no / a short workspace must not touch it and still give the gradient
    (the LDS-atomic plane kernel runs)."""
    from r3det import _C
    L = _C.lib()
    N, C, H, W, stride = 1, 512, 64, 64, 16
    r = np.random.default_rng(3)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 4)
    with O.twin():
        want = O.fr_backward(top[:, :4], boxes, 1 / stride, 1)
    tol = 1e-5 * max(1.0, np.abs(want).max())
    t, b = dev(top), dev(boxes)
    need = int(L.r3det_fr_backward_workspace_bytes(N, H, W, 1))
    assert need > 0 and int(L.r3det_fr_backward_workspace_bytes(N, 32, 32, 1)) > 0
    assert int(L.r3det_fr_backward_workspace_bytes(N, H, W, 5)) > need
    assert int(L.r3det_fr_backward_workspace_bytes(N, 256, 256, 1)) == 0      # a plane beyond the LDS
    assert int(L.r3det_fr_backward_workspace_bytes(N, H, W, 3)) == 0
    guard = torch.full((need,), 7, dtype=torch.uint8, device='cuda')
    for ws_ptr, ws_bytes in ((None, 0), (_C.ptr(guard), need - 16)):
        g = torch.full((N, C, H, W), float('nan'), device='cuda')
        _C.check(L.r3det_feature_refine_backward_ws(_C.ptr(t), _C.ptr(b), N, C, H, W, 1 / stride, 1, _C.ptr(g), 1,
                                                    ws_ptr, ws_bytes, _C.stream()), "bwd")
        assert np.abs(g[:, :4].cpu().numpy() - want).max() <= tol
    assert bool((guard == 7).all())


@pytest.mark.parametrize("shape", [(2, 512, 128, 128, 8), (2, 512, 64, 64, 16), (1, 8, 64, 64, 16), (2, 16, 32, 32, 32),
                                   (1, 6, 100, 167, 8), (1, 3, 150, 200, 8), (1, 2, 256, 256, 8)])
@pytest.mark.parametrize("points", [1, 5])
def test_autograd_backward_uses_the_index_from_forward(shape, points):
    """feature_refine(x.requires_grad) builds the backward's index of the boxes in the forward pass (split form:
    r3det_feature_refine_backward_index / _indexed); the backward is the gather alone.  Planes beyond the gather's
    limits fall back inside the same function.  Gradient against the oracle either way; the C-ABI calls directly:
    the split form == the one-call form, bit for bit, and argument errors are refused."""
    from r3det import _C
    from r3det.ops.feature_refine import feature_refine, fr_backward, fr_backward_index, fr_backward_indexed
    N, C, H, W, stride = shape
    r = np.random.default_rng(5)
    x = dev(r.normal(size=(N, C, H, W)).astype(np.float32)).requires_grad_(True)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 9)
    y = feature_refine(x, dev(boxes), 1 / stride, points)
    assert (y.grad_fn.index is not None) == (H * W <= 32768)
    y.backward(dev(top))
    nc = min(C, 4)
    with O.twin():
        want = O.fr_backward(top[:, :nc], boxes, 1 / stride, points)
    assert np.abs(x.grad[:, :nc].cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    # no gradient wanted: no index
    y2 = feature_refine(x.detach(), dev(boxes), 1 / stride, points)
    assert y2.grad_fn is None and torch.equal(y2, y.detach())
    if H * W > 32768:
        assert fr_backward_index(dev(boxes), N, C, H, W, 1 / stride, points) is None
        return
    index = fr_backward_index(dev(boxes), N, C, H, W, 1 / stride, points)
    g1 = torch.full((N, C, H, W), float('nan'), device='cuda')
    assert fr_backward_indexed(dev(top), points, g1, index)
    g2 = torch.full((N, C, H, W), float('nan'), device='cuda')
    fr_backward(dev(top), dev(boxes), 1 / stride, points, g2, overwrite=True)
    assert torch.equal(g1, g2) and torch.equal(g1, x.grad)
    L = _C.lib()
    t = dev(top)
    args = (_C.ptr(t), N, C, H, W, points, _C.ptr(g1), 1)
    assert L.r3det_feature_refine_backward_indexed(*args, _C.ptr(index), index.numel() - 16, _C.stream()) == -3
    assert L.r3det_feature_refine_backward_indexed(*args, None, 0, _C.stream()) == -1
    assert L.r3det_feature_refine_backward_indexed(_C.ptr(t), N, C, H, W, 3, _C.ptr(g1), 1, _C.ptr(index),
                                                   index.numel(), _C.stream()) == -1


@pytest.mark.parametrize("points", [1, 5])
@pytest.mark.parametrize("NC", [(2, 16), (1, 6)])
def test_levels_node_equals_per_level_nodes(points, NC):
    """feature_refine_levels = one autograd node for the five pyramid levels (one library call each for the samplers,
    the backward's indexes, the gathers): outputs and gradients bit-identical to one feature_refine node per level;
    the C ABI's levels calls refuse a short workspace."""
    import ctypes
    from r3det import _C, synthetic as syn
    from r3det.ops.feature_refine import feature_refine, feature_refine_levels
    N, C = NC
    feats, boxes = syn.fr_pyramid(N, C, 5, device='cuda')
    scales = [1.0 / s for s in syn.STRIDES]
    gs = [torch.randn_like(f) for f in feats]
    xs = [f.clone().requires_grad_(True) for f in feats]
    outs = feature_refine_levels(xs, boxes, scales, points)
    torch.autograd.backward(outs, gs)
    for f, b, s, o, x, g in zip(feats, boxes, scales, outs, xs, gs):
        x1 = f.clone().requires_grad_(True)
        o1 = feature_refine(x1, b, s, points)
        o1.backward(g)
        assert torch.equal(o, o1) and torch.equal(x.grad, x1.grad)
    # the indexes of the levels from one grouped launch (default) == level by level (option frb_impl 6)
    _C.set_option("frb_impl", 6)
    xs6 = [f.clone().requires_grad_(True) for f in feats]
    torch.autograd.backward(feature_refine_levels(xs6, boxes, scales, points), gs)
    _C.set_option("frb_impl", 0)
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(xs6, xs))
    # gradients only for some levels / none
    xs2 = [f.clone().requires_grad_(i % 2 == 0) for i, f in enumerate(feats)]
    outs2 = feature_refine_levels(xs2, boxes, scales, points)
    torch.autograd.backward([o for i, o in enumerate(outs2) if i % 2 == 0], [g for i, g in enumerate(gs) if i % 2 == 0])
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(xs2[::2], xs[::2])) and xs2[1].grad is None
    assert all(o.grad_fn is None for o in feature_refine_levels(feats, boxes, scales, points))
    L = _C.lib()
    n = len(feats)
    arr_i, arr_p = ctypes.c_int * n, ctypes.c_void_p * n
    H, W = arr_i(*[f.size(2) for f in feats]), arr_i(*[f.size(3) for f in feats])
    sc = (ctypes.c_float * n)(*scales)
    need = int(L.r3det_fr_backward_levels_workspace_bytes(n, N, H, W, points))
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device='cuda')
    bp = arr_p(*[b.data_ptr() for b in boxes])
    assert L.r3det_feature_refine_backward_index_levels(n, bp, N, C, H, W, sc, points, _C.ptr(ws), need - 256,
                                                        _C.stream()) == -3
    assert L.r3det_feature_refine_backward_index_levels(n, None, N, C, H, W, sc, points, _C.ptr(ws), need,
                                                        _C.stream()) == -1


@pytest.mark.parametrize("points", [1, 5])
@pytest.mark.parametrize("NC", [(2, 16), (2, 256)])
def test_levels_node_channels_last_equals_per_level_nodes(points, NC):
    """feature_refine_levels on channels_last inputs: the _nhwc forms of the three levels calls (samplers; the indexes
    of all levels from one grouped launch; the coarse levels' gathers one grid) -- outputs and gradients bit-identical
    to one feature_refine node per level, stay channels_last, and equal one launch per level (option frb_impl 6);
    overwrite and accumulate mode of the C ABI's gather call."""
    import ctypes
    from r3det import _C, synthetic as syn
    from r3det.ops.feature_refine import feature_refine, feature_refine_levels
    N, C = NC
    cl = torch.channels_last
    feats, boxes = syn.fr_pyramid(N, C, 7, device='cuda')
    feats = [f.contiguous(memory_format=cl) for f in feats]
    scales = [1.0 / s for s in syn.STRIDES]
    gs = [torch.randn_like(f) for f in feats]
    xs = [f.clone().requires_grad_(True) for f in feats]
    outs = feature_refine_levels(xs, boxes, scales, points)
    torch.autograd.backward(outs, gs)
    for f, b, s, o, x, g in zip(feats, boxes, scales, outs, xs, gs):
        x1 = f.clone().requires_grad_(True)
        o1 = feature_refine(x1, b, s, points)
        o1.backward(g)
        if f.size(2) > 1:  # (1 x 1 maps are both layouts at once)
            assert o.is_contiguous(memory_format=cl) and x.grad.is_contiguous(memory_format=cl)
        assert torch.equal(o, o1) and torch.equal(x.grad, x1.grad)
    _C.set_option("frb_impl", 6)
    try:
        xs6 = [f.clone().requires_grad_(True) for f in feats]
        torch.autograd.backward(feature_refine_levels(xs6, boxes, scales, points), gs)
    finally:
        _C.set_option("frb_impl", 0)
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(xs6, xs))
    # the C ABI: accumulate mode, short workspace
    L = _C.lib()
    n = len(feats)
    arr_i, arr_p = ctypes.c_int * n, ctypes.c_void_p * n
    H, W = arr_i(*[f.size(2) for f in feats]), arr_i(*[f.size(3) for f in feats])
    sc = (ctypes.c_float * n)(*scales)
    need = int(L.r3det_fr_backward_nhwc_levels_workspace_bytes(n, N, H, W, points))
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device='cuda')
    bp = arr_p(*[b.data_ptr() for b in boxes])
    assert L.r3det_feature_refine_backward_nhwc_index_levels(n, bp, N, H, W, sc, points, _C.ptr(ws), need - 256,
                                                             _C.stream()) == -3
    assert L.r3det_feature_refine_backward_nhwc_index_levels(n, bp, N, H, W, sc, points, _C.ptr(ws), need, _C.stream()) == 0
    pre = [torch.randn_like(f) for f in feats]
    acc = [p.clone(memory_format=torch.preserve_format) for p in pre]
    assert L.r3det_feature_refine_backward_nhwc_levels_indexed(
        n, arr_p(*[g.data_ptr() for g in gs]), N, C, H, W, points, arr_p(*[a.data_ptr() for a in acc]), 0, _C.ptr(ws),
        need, _C.stream()) == 0
    for a, p, x in zip(acc, pre, xs):
        assert (a - (p + x.grad)).abs().max().item() <= 1e-5 * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize("NC", [(2, 16), (4, 256)])
def test_backward_levels_equals_per_level_calls_in_both_modes(NC):
    """r3det_feature_refine_backward_index_levels + _backward_levels_indexed (the coarse levels' gathers one grid) =
    one r3det_feature_refine_backward_ws call per level, bit for bit, writing and ACCUMULATING into bottom_grad
    (feature_refine_cuda.cpp:44-66 accumulates), and with one launch per level (option frb_impl 6)."""
    from r3det import _C, synthetic as syn
    from r3det.ops.feature_refine import fr_backward, fr_backward_levels
    N, C = NC
    feats, boxes = syn.fr_pyramid(N, C, 11, device='cuda')
    scales = [1.0 / s for s in syn.STRIDES]
    gs = [torch.randn_like(f) for f in feats]
    pre = [torch.randn_like(f) for f in feats]
    for overwrite in (True, False):
        want = [p.clone() for p in pre]
        for g, b, s, o in zip(gs, boxes, scales, want):
            fr_backward(g, b, s, 1, o, overwrite=overwrite)
        for impl in (0, 6):
            got = [p.clone() for p in pre]
            _C.set_option("frb_impl", impl)
            try:
                fr_backward_levels(gs, boxes, scales, 1, got, overwrite=overwrite)
            finally:
                _C.set_option("frb_impl", 0)
            for lvl, (a, w) in enumerate(zip(got, want)):
                assert torch.equal(a, w), (overwrite, impl, lvl)


@pytest.mark.parametrize("points", [1, 5])
def test_forward_levels_equals_per_level_calls(points):
    """r3det_feature_refine_forward_levels = the module's per-level loop: bit-identical outputs."""
    from r3det import synthetic as syn
    from r3det.ops.feature_refine import fr_forward, fr_forward_levels
    N, C = 2, 16
    feats, boxes = syn.fr_pyramid(N, C, 3, device='cuda')
    scales = [1.0 / s for s in syn.STRIDES]
    outs = [torch.full_like(f, float('nan')) for f in feats]
    fr_forward_levels(feats, boxes, scales, points, outs)
    for f, b, s, o in zip(feats, boxes, scales, outs):
        want = torch.empty_like(f)
        fr_forward(f, b, s, points, want)
        assert torch.equal(o, want)
    with pytest.raises(RuntimeError):
        fr_forward_levels(feats, boxes[::-1], scales, points, outs)


@pytest.mark.parametrize("form", ["auto", "wide", "pairs"])
def test_module_nhwc_bench_shape_directly_against_the_oracle(form):
    """The roofline launch itself -- r3det_feature_refine_module_nhwc at (4, 256, 128, 128), the bench's box field --
    against the ORACLE on a sample of channels of every image (not through the NCHW kernel): P = (a + bias_a) +
    (b + bias_b) in fp32 on the host, out = residual + (P + sample(P)), bit for bit."""
    from r3det import _C, synthetic as syn
    from r3det.ops.feature_refine import fr_module_nhwc
    N, C, H, W, stride = 4, 256, 128, 128, 8
    cl = torch.channels_last
    g = torch.Generator(device='cuda').manual_seed(5)
    mk = lambda: torch.randn(N, C, H, W, device='cuda', generator=g).contiguous(memory_format=cl)  # noqa: E731
    a, b, r = mk(), mk(), mk()
    ba, bb = torch.randn(C, device='cuda', generator=g), torch.randn(C, device='cuda', generator=g)
    boxes = syn.fr_level_boxes(N, H, W, stride, 3, device='cuda')
    out = torch.full_like(a, float('nan'))
    _C.set_option("fr_dbg", {"auto": 0, "wide": 8, "pairs": 9}[form])
    try:
        assert fr_module_nhwc(a, b, ba, bb, r, boxes, 1 / stride, 1, out)
    finally:
        _C.set_option("fr_dbg", 0)
    chans = [0, 3, 64, 129, 255]
    an, bn, rn = (t[:, chans].cpu().numpy() for t in (a, b, r))
    ban, bbn = ba[chans].cpu().numpy()[None, :, None, None], bb[chans].cpu().numpy()[None, :, None, None]
    P = (an + ban) + (bn + bbn)
    with O.twin():
        want = rn + O.fr_forward(np.ascontiguousarray(P), boxes.cpu().numpy(), 1 / stride, 1, threads=8)
    assert np.array_equal(out[:, chans].cpu().numpy(), want)


@pytest.mark.parametrize("NC", [(4, 256), (2, 64), (1, 12)])
@pytest.mark.parametrize("adversarial", [False, True])
def test_nhwc_levels_calls_equal_per_level_calls(NC, adversarial):
    """r3det_feature_refine_forward_levels_nhwc / _module_levels_nhwc (level 0 in the wide form alone, the coarse levels
    as ONE grid over their tile pairs) == one r3det_feature_refine_forward_nhwc / _module_nhwc call per level, bit for
    bit; points = 5 (no grouped form) goes level by level inside the same call."""
    from r3det import synthetic as syn
    from r3det.ops.feature_refine import (fr_forward_levels_nhwc, fr_forward_nhwc, fr_module_levels_nhwc,
                                          fr_module_nhwc)
    N, C = NC
    cl = torch.channels_last
    feats, boxes = syn.fr_pyramid(N, C, 7, adversarial=adversarial, device='cuda')
    g = torch.Generator(device='cuda').manual_seed(11)
    mk = lambda f: torch.randn(f.shape, device='cuda', generator=g).contiguous(memory_format=cl)  # noqa: E731
    a, b, r = [mk(f) for f in feats], [mk(f) for f in feats], [f.contiguous(memory_format=cl) for f in feats]
    ba, bb = torch.randn(C, device='cuda', generator=g), torch.randn(C, device='cuda', generator=g)
    scales = [1.0 / s for s in syn.STRIDES]
    for points in (1, 5):
        outs = [torch.full_like(t, float('nan')) for t in a]
        assert fr_forward_levels_nhwc(a, boxes, scales, points, outs)
        for t, bx, sc, o in zip(a, boxes, scales, outs):
            want = torch.empty_like(t)
            assert fr_forward_nhwc(t, bx, sc, points, want)
            assert torch.equal(o, want)
        for cb in (b, None):
            outs = [torch.full_like(t, float('nan')) for t in a]
            assert fr_module_levels_nhwc(a, cb, ba, bb, r, boxes, scales, points, outs)
            for i, (t, rr, bx, sc, o) in enumerate(zip(a, r, boxes, scales, outs)):
                want = torch.empty_like(t)
                assert fr_module_nhwc(t, cb[i] if cb is not None else None, ba, bb, rr, bx, sc, points, want)
                assert torch.equal(o, want)
    with pytest.raises(RuntimeError):
        fr_forward_levels_nhwc(a, boxes[::-1], scales, 1, outs)


@pytest.mark.parametrize("shape", [(2, 512, 128, 128, 8), (1, 1024, 128, 128, 8), (3, 512, 64, 64, 16)])
@pytest.mark.parametrize("adversarial", [False, True])
def test_module_fused_sampler_bit_identical(shape, adversarial):
    """r3det_feature_refine_module_prepared = residual + fr(a + b) of the module
    (feature_refine_module.py:121-126) in one launch: bit-identical to the three it replaces."""
    from r3det.ops.feature_refine import fr_forward, fr_module_prepared, fr_prepare
    N, C, H, W, stride = shape
    g = torch.Generator(device='cuda').manual_seed(N * 7 + H)
    a, b, res = (torch.randn(N, C, H, W, device='cuda', generator=g) for _ in range(3))
    boxes = dev(fr_boxes(N, H, W, stride, 13, adversarial=adversarial))
    table = fr_prepare(boxes, N, H, W, 1 / stride)
    out = torch.full_like(a, float('nan'))
    assert fr_module_prepared(a, b, res, table, out)
    mixed = a + b
    sampled = torch.empty_like(mixed)
    fr_forward(mixed, boxes, 1 / stride, 1, sampled)
    assert torch.equal(out, res + sampled)
    # shapes / channel counts without the cell kernel are refused, nothing is written
    small = torch.randn(1, 4, H, W, device='cuda')
    o2 = torch.full_like(small, 5.0)
    t2 = fr_prepare(boxes[:H * W], 1, H, W, 1 / stride)
    assert not fr_module_prepared(small, small, small, t2, o2) and bool((o2 == 5.0).all())


@pytest.mark.parametrize("shape", [(2, 256, 64, 64), (1, 100, 13, 7), (3, 64, 1, 1), (1, 65, 128, 128)])
@pytest.mark.parametrize("two", [True, False])
def test_mix_to_nchw(shape, two):
    """r3det_frm_mix_nchw: (a + bias_a) + (b + bias_b) of channels_last inputs, written NCHW; exact."""
    from r3det.ops.epilogue import mix_to_nchw
    N, C, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(C + H)
    a = torch.randn(N, C, H, W, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    b = torch.randn(N, C, H, W, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    ba, bb = torch.randn(C, device='cuda', generator=g), torch.randn(C, device='cuda', generator=g)
    if a.is_contiguous():  # (H = W = 1: both layouts at once -> not taken)
        assert mix_to_nchw(a, b, ba, bb) is None
        return
    if two:
        got = mix_to_nchw(a, b, ba, bb)
        want = (a + ba.view(1, -1, 1, 1)) + (b + bb.view(1, -1, 1, 1))
    else:
        got = mix_to_nchw(a)
        want = a
    assert got.is_contiguous() and torch.equal(got, want.contiguous())
    assert mix_to_nchw(a.contiguous()) is None  # NCHW input: nothing to do here


def test_module_channels_last_inference_path():
    """FeatureRefineModule on channels_last features without grad: raw convolutions + ONE channels_last launch
    for the whole tail (r3det_feature_refine_module_nhwc), output channels_last.  Against the module's own
    three-step NCHW form on the same weights (different convolution kernels per layout: not bitwise), and against
    the NCHW sampler applied to the channels_last convolutions' outputs."""
    from r3det.ops import FeatureRefineModule
    from r3det.ops import feature_refine as frmod
    from r3det import synthetic as syn
    torch.manual_seed(3)
    N, C = 2, 256
    feats, boxes = syn.fr_pyramid(N, C, 5, device='cuda')
    m = FeatureRefineModule(C, list(syn.STRIDES)).cuda()
    for conv in (m.conv_5_1, m.conv_1_5, m.conv_1_1):
        torch.nn.init.normal_(conv.weight, 0, 0.05)
        torch.nn.init.normal_(conv.bias, 0, 0.5)
    rois = [[b.view(N, -1, 5)[i] for b in boxes] for i in range(N)]
    with torch.no_grad():
        want = m(feats, rois)                         # NCHW module, NCHW features
        mcl = m.to(memory_format=torch.channels_last)
        cl_feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
        frmod.NHWC_ONLY = True
        try:
            got = mcl(cl_feats, rois)
        finally:
            frmod.NHWC_ONLY = False
        for l, (g, w, f) in enumerate(zip(got, want, cl_feats)):
            assert g.shape == w.shape
            assert g.is_contiguous(memory_format=torch.channels_last)
            assert torch.allclose(g, w, rtol=1e-4, atol=1e-4), float((g - w).abs().max())
            # the same arithmetic through the NCHW sampler on the channels_last convolutions' outputs (recomputed here:
            # MIOpen may pick another kernel for the repeated call, so rounding-level, not bitwise; the bitwise
            # statement is test_module_nhwc_bit_identical_to_the_nchw_steps)
            a = torch.nn.functional.conv2d(mcl.conv_1_5(f), mcl.conv_5_1.weight, None, 1, mcl.conv_5_1.padding)
            b = torch.nn.functional.conv2d(f, mcl.conv_1_1.weight, None)
            mixed = ((a + mcl.conv_5_1.bias.view(1, -1, 1, 1)) + (b + mcl.conv_1_1.bias.view(1, -1, 1, 1))).contiguous()
            sampled = torch.empty_like(mixed)
            frmod.fr_forward(mixed, boxes[l], 1.0 / syn.STRIDES[l], 1, sampled)
            assert torch.allclose(g.contiguous(), f.contiguous() + sampled, rtol=1e-5, atol=1e-5), l


# (the last: 1024 workgroups of the wide form = the library's own choice for that shape)
NHWC_SHAPES = [(n, (c + 3) // 4 * 4, h, w, st) for n, c, h, w, st in SHAPES] + [(2, 256, 64, 64, 16), (1, 260, 9, 5, 32),
                                                                                (4, 32, 128, 128, 8)]


@pytest.mark.parametrize("shape", NHWC_SHAPES)
@pytest.mark.parametrize("points", [1, 5])
@pytest.mark.parametrize("adversarial", [False, True])
@pytest.mark.parametrize("form", ["auto", "wide"])
def test_forward_nhwc_bit_exact(shape, points, adversarial, form):
    """The channels_last sampler against the twin oracle on the transposed input: every spatial shape of SHAPES with C
    rounded up to a multiple of 4, bit-exact.  form "wide": option fr_dbg 8 forces the wide regions form on the square
    maps with a side that is a multiple of 8 (the library takes it by itself from 512 workgroups)."""
    from r3det import _C
    N, C, H, W, stride = shape
    if form == "wide" and (points != 1 or H != W or H % 8):
        pytest.skip("the wide form does not take this shape")
    _C.set_option("fr_dbg", 8 if form == "wide" else 0)
    try:
        _forward_nhwc_case(shape, points, adversarial)
    finally:
        _C.set_option("fr_dbg", 0)


def _forward_nhwc_case(shape, points, adversarial):
    from r3det.ops.feature_refine import fr_forward_nhwc
    N, C, H, W, stride = shape
    r = np.random.default_rng(5 + C)
    cs = min(C, 24)  # the oracle is serial: check the first and the last channels of wide maps
    feat = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 6, adversarial=adversarial)
    sel = np.r_[0:cs // 2, C - cs // 2:C]
    with O.twin():
        want = O.fr_forward(np.ascontiguousarray(feat[:, sel]), boxes, 1 / stride, points, threads=8)
    x = dev(feat).contiguous(memory_format=torch.channels_last)
    out = torch.full((N, C, H, W), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    assert fr_forward_nhwc(x, dev(boxes), 1 / stride, points, out)
    assert not bool(torch.isnan(out).any())
    assert np.array_equal(out[:, torch.from_numpy(sel).cuda()].contiguous().cpu().numpy(), want)


def test_forward_nhwc_refuses_odd_channel_counts():
    from r3det.ops.feature_refine import fr_forward_nhwc
    x = torch.randn(1, 6, 8, 8, device='cuda').contiguous(memory_format=torch.channels_last)
    out = torch.full_like(x, 3.0)
    assert not fr_forward_nhwc(x, dev(fr_boxes(1, 8, 8, 8, 1)), 0.125, 1, out)
    assert bool((out == 3.0).all())
    with pytest.raises(RuntimeError):
        fr_forward_nhwc(x.contiguous(), dev(fr_boxes(1, 8, 8, 8, 1)), 0.125, 1, out)  # NCHW memory


@pytest.mark.parametrize("shape", [(2, 256, 128, 128, 8), (4, 256, 64, 64, 16), (3, 64, 13, 7, 32), (1, 256, 1, 1, 128),
                                   (4, 256, 128, 128, 8),   # bench.py's roofline launch: 2048 workgroups, XCD band remap on
                                   (8, 256, 128, 128, 8), (1, 256, 128, 128, 8), (3, 256, 32, 32, 32), (2, 64, 8, 8, 128),
                                   (1, 512, 16, 16, 64)])
@pytest.mark.parametrize("points", [1, 5])
@pytest.mark.parametrize("with_b", [True, False])
@pytest.mark.parametrize("form", ["pairs", "wide", "auto"])
def test_module_nhwc_bit_identical_to_the_nchw_steps(shape, points, with_b, form):
    """r3det_feature_refine_module_nhwc = residual + fr((a + bias_a) + (b + bias_b)) in one launch on
    channels_last memory: bit-identical to the elementwise steps + the NCHW sampler (itself pinned to the
    oracle above).  form "wide" (option fr_dbg 8): the 4 x 8 / 8 x 4 / 8 x 8 regions form for square maps with a side
    that is a multiple of 8; "pairs" (fr_dbg 9): the 4 x 4 tile pairs; "auto": the library's choice (wide from 512
    workgroups)."""
    from r3det import _C
    from r3det.ops.feature_refine import fr_forward, fr_module_nhwc
    N, C, H, W, stride = shape
    if form != "auto" and (points != 1 or H != W or H % 8):
        pytest.skip("this shape takes the same launch either way")
    _C.set_option("fr_dbg", {"pairs": 9, "wide": 8, "auto": 0}[form])
    try:
        _module_nhwc_case(shape, points, with_b)
    finally:
        _C.set_option("fr_dbg", 0)


def _module_nhwc_case(shape, points, with_b):
    from r3det.ops.feature_refine import fr_forward, fr_module_nhwc
    N, C, H, W, stride = shape
    g = torch.Generator(device='cuda').manual_seed(C + H + points)
    cl = torch.channels_last
    a, b, res = (torch.randn(N, C, H, W, device='cuda', generator=g).contiguous(memory_format=cl) for _ in range(3))
    ba, bb = torch.randn(C, device='cuda', generator=g), torch.randn(C, device='cuda', generator=g)
    boxes = dev(fr_boxes(N, H, W, stride, 17, adversarial=True))
    out = torch.full_like(a, float('nan'))
    assert fr_module_nhwc(a, b if with_b else None, ba, bb if with_b else None, res, boxes, 1 / stride, points, out)
    mixed = a + ba.view(1, -1, 1, 1)
    if with_b:
        mixed = mixed + (b + bb.view(1, -1, 1, 1))
    mixed = mixed.contiguous()
    sampled = torch.empty_like(mixed)
    fr_forward(mixed, boxes, 1 / stride, points, sampled)
    assert torch.equal(out.contiguous(), res.contiguous() + sampled)


@pytest.mark.parametrize("level", [0, 1, 2, 4])
def test_full_size_properties(level):
    """BASELINE shapes (N = 4, C = 256, the pyramid of a 1024^2 input), where the oracle is too slow:
    size-independent properties.  (1) every channel sees the same taps: a plane replicated over the
    channels gives identical output channels, and equals the first 2 channels' oracle result;
    (2) linearity of the forward; (3) the backward is the forward's adjoint: <fr(x), g> = <x, fr^T(g)>."""
    from r3det import synthetic as syn
    from r3det.ops.feature_refine import fr_backward, fr_forward
    N, C = 4, 256
    feats, boxes = syn.fr_pyramid(N, C, 21, device='cuda')
    x, b, s = feats[level], boxes[level], 1.0 / syn.STRIDES[level]
    g = torch.Generator(device='cuda').manual_seed(level)
    y = torch.randn(x.shape, device='cuda', generator=g)
    fx, fy, fxy = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    fr_forward(x, b, s, 1, fx)
    fr_forward(y, b, s, 1, fy)
    # (1)
    rep = x[:, :1].expand_as(x).contiguous()
    frep = torch.empty_like(x)
    fr_forward(rep, b, s, 1, frep)
    assert torch.equal(frep, frep[:, :1].expand_as(frep))
    assert torch.equal(frep[:, 0], fx[:, 0])
    with O.twin():
        want = O.fr_forward(x[:1, :2].cpu().numpy(), b.view(N, -1, 5)[0].cpu().numpy(), s, 1)
    assert np.array_equal(fx[:1, :2].cpu().numpy(), want)
    # (2)
    fr_forward(0.5 * x - 2.0 * y, b, s, 1, fxy)
    assert torch.allclose(fxy, 0.5 * fx - 2.0 * fy, rtol=1e-5, atol=1e-5)
    # (3)
    gt = torch.randn(x.shape, device='cuda', generator=g)
    back = torch.empty_like(x)
    fr_backward(gt, b, s, 1, back, overwrite=True)
    lhs = (fx.double() * gt.double()).sum().item()
    rhs = (x.double() * back.double()).sum().item()
    scale = (fx.double().abs() * gt.double().abs()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * scale, (lhs, rhs, scale)
