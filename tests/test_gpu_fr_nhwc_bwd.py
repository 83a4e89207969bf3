"""GPU parity: the channels_last backward of the FR sampler (r3det_feature_refine_backward_nhwc: inverse tap index +
gather, csrc/r3_frb.hip) against the oracle's restatement of feature_refine_backward_kernel
(feature_refine_kernel.cu:165-230) and against the NCHW backward of this library.  The reference sums float atomics
in any order => tolerance 1e-5 relative to the gradient scale (written below); the gather itself sums in ONE order,
so two runs must agree bit for bit."""
import numpy as np
import pytest
import torch

from helpers import fr_boxes
from oracle import api as O
from test_gpu_fr import NHWC_SHAPES, SHAPES, dev

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def cl(a):
    return dev(a).contiguous(memory_format=CL)


@pytest.fixture(params=[0, 1, 2], ids=["auto", "general-index", "unpaired-gather"])
def frb_impl(request):
    """0 = automatic (points = 1: bands sorted in LDS; square tile grids: transposed tile pairs sharing rows through
    LDS), 1 = the general index form (LDS counters + fill + per-list sort) whatever the shape, 2 = one tile per
    workgroup in the gather."""
    from r3det import _C
    _C.set_option("frb_impl", request.param)
    yield request.param
    _C.set_option("frb_impl", 0)


@pytest.mark.parametrize("shape", NHWC_SHAPES)
@pytest.mark.parametrize("points", [1, 5])
@pytest.mark.parametrize("adversarial", [False, True])
def test_backward_nhwc_vs_oracle_and_nchw(shape, points, adversarial, frb_impl):
    """Every SHAPES entry (C rounded up to a multiple of 4): overwrite mode against the oracle on the first / last
    channels, against the NCHW backward on all channels, accumulate mode, and run-to-run determinism."""
    from r3det.ops.feature_refine import fr_backward, fr_backward_nhwc
    N, C, H, W, stride = shape
    r = np.random.default_rng(31 + C)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 8, adversarial=adversarial)
    cs = min(C, 8)
    sel = np.r_[0:cs // 2, C - cs // 2:C]
    with O.twin():
        want = O.fr_backward(np.ascontiguousarray(top[:, sel]), boxes, 1 / stride, points)
    tol = 1e-5 * max(1.0, np.abs(want).max())
    t, b = cl(top), dev(boxes)
    g = torch.full((N, C, H, W), float('nan'), device='cuda').contiguous(memory_format=CL)
    assert fr_backward_nhwc(t, b, 1 / stride, points, g, overwrite=True)
    assert not bool(torch.isnan(g).any())
    got = g[:, torch.from_numpy(sel).cuda()].contiguous().cpu().numpy()
    assert np.abs(got - want).max() <= tol
    # the NCHW backward of the library (atomics or packed: any order) on all channels
    g2 = torch.empty((N, C, H, W), device='cuda')
    fr_backward(dev(top), b, 1 / stride, points, g2, overwrite=True)
    assert (g.contiguous() - g2).abs().max().item() <= 1e-5 * max(1.0, g2.abs().max().item())
    # one summation order: bit-identical run to run
    g3 = torch.empty_like(g)
    assert fr_backward_nhwc(t, b, 1 / stride, points, g3, overwrite=True)
    assert torch.equal(g, g3)
    # the reference's calling convention: accumulate into the caller's buffer
    acc = g.clone()
    assert fr_backward_nhwc(t, b, 1 / stride, points, acc, overwrite=False)
    assert (acc - 2 * g).abs().max().item() <= 2e-5 * max(1.0, g.abs().max().item())


@pytest.mark.parametrize("shape", [(2, 8, 128, 128, 8), (2, 8, 64, 64, 16), (1, 12, 20, 36, 16)])
@pytest.mark.parametrize("pile", [2, 17, 49, 300, 2100, -1])
def test_backward_nhwc_piles(shape, pile, frb_impl):
    """`pile` positions of the last image sample one cell (-1: every position of it): lists longer than the wave
    (64 entries per pass), longer than the general form's sorted range, and -- 2100 sources = 8400 entries in one
    band, or all of them -- more than the sort form's LDS list holds (that band then takes the general form inside
    the same launch); out-of-range samples contribute nothing."""
    from r3det.ops.feature_refine import fr_backward_nhwc
    N, C, H, W, stride = shape
    r = np.random.default_rng(pile + 200)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 21).reshape(N, H * W, 5)
    idx = np.arange(H * W) if pile < 0 else r.choice(H * W, min(pile, H * W), replace=False)
    boxes[-1, idx, 0] = (H // 3 + 0.3) * stride + r.uniform(0, 0.6 * stride, idx.size).astype(np.float32)
    boxes[-1, idx, 1] = (W // 2 + 0.6) * stride
    boxes[0, :5, 0] = -40.0 * stride   # out of range: no contribution (feature_refine_kernel.cu:72-79)
    boxes[0, 5:9, 1] = (W + 3.0) * stride
    boxes = boxes.reshape(-1, 5)
    with O.twin():
        want = O.fr_backward(top, boxes, 1 / stride, 1)
    g = torch.full((N, C, H, W), float('nan'), device='cuda').contiguous(memory_format=CL)
    assert fr_backward_nhwc(cl(top), dev(boxes), 1 / stride, 1, g, overwrite=True)
    assert np.abs(g.contiguous().cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_backward_nhwc_split_form_and_argument_errors():
    from r3det import _C
    from r3det.ops.feature_refine import fr_backward_nhwc, fr_backward_nhwc_index
    L = _C.lib()
    N, C, H, W, stride = 2, 16, 24, 40, 8
    r = np.random.default_rng(2)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = dev(fr_boxes(N, H, W, stride, 3, adversarial=True))
    g1 = torch.empty((N, C, H, W), device='cuda').contiguous(memory_format=CL)
    g2 = torch.empty_like(g1)
    assert fr_backward_nhwc(cl(top), boxes, 1 / stride, 1, g1, overwrite=True)
    index = fr_backward_nhwc_index(boxes, N, H, W, 1 / stride, 1)
    assert index is not None
    assert fr_backward_nhwc(cl(top), None, 1 / stride, 1, g2, overwrite=True, index=index)
    assert torch.equal(g1, g2)
    # shapes / arguments the entry does not take: nothing is launched
    assert int(L.r3det_fr_backward_nhwc_workspace_bytes(1, 8, 5000, 1)) == 0
    assert int(L.r3det_fr_backward_nhwc_workspace_bytes(1, 8, 8, 3)) == 0
    need = int(L.r3det_fr_backward_nhwc_workspace_bytes(N, H, W, 1))
    assert need >= N * H * W * (8 + 4 * 8)
    ws = torch.empty(need, dtype=torch.uint8, device='cuda')
    t = cl(top)
    args = (_C.ptr(t), _C.ptr(boxes), N, C, H, W, 1 / stride, 1, _C.ptr(g1), 1)
    assert L.r3det_feature_refine_backward_nhwc(*args, _C.ptr(ws), need - 1, _C.stream()) == -3
    assert L.r3det_feature_refine_backward_nhwc(*args, None, 0, _C.stream()) == -1
    assert L.r3det_feature_refine_backward_nhwc(_C.ptr(t), _C.ptr(boxes), N, 6, H, W, 1 / stride, 1, _C.ptr(g1), 1,
                                                _C.ptr(ws), need, _C.stream()) == -1  # C % 4
    assert L.r3det_feature_refine_backward_nhwc(_C.ptr(t), _C.ptr(boxes), N, C, H, W, 1 / stride, 2, _C.ptr(g1), 1,
                                                _C.ptr(ws), need, _C.stream()) == -1  # points
    odd = torch.randn(1, 6, 8, 8, device='cuda').contiguous(memory_format=CL)
    assert not fr_backward_nhwc(odd, dev(fr_boxes(1, 8, 8, 8, 1)), 0.125, 1, torch.empty_like(odd), overwrite=True)
    with pytest.raises(RuntimeError):
        fr_backward_nhwc(odd.contiguous(), dev(fr_boxes(1, 8, 8, 8, 1)), 0.125, 1, torch.empty_like(odd))  # NCHW memory


@pytest.mark.parametrize("level", [0, 1, 3])
def test_full_size_adjoint_nhwc(level):
    """BASELINE shapes (N = 4, C = 256, pyramid of a 1024^2 input): the NHWC backward is the adjoint of the NHWC
    forward, <fr(x), g> = <x, fr^T(g)>, and equals the NCHW backward within the summation-order tolerance."""
    from r3det import synthetic as syn
    from r3det.ops.feature_refine import fr_backward, fr_backward_nhwc, fr_forward_nhwc
    N, C = 4, 256
    feats, boxes = syn.fr_pyramid(N, C, 21, device='cuda')
    x, b, s = feats[level].contiguous(memory_format=CL), boxes[level], 1.0 / syn.STRIDES[level]
    gen = torch.Generator(device='cuda').manual_seed(level)
    g = torch.randn(x.shape, device='cuda', generator=gen).contiguous(memory_format=CL)
    fx, bg = torch.empty_like(x), torch.empty_like(x)
    assert fr_forward_nhwc(x, b, s, 1, fx)
    assert fr_backward_nhwc(g, b, s, 1, bg, overwrite=True)
    lhs = (fx.double() * g.double()).sum().item()
    rhs = (x.double() * bg.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0)
    ref = torch.empty(x.shape, device='cuda')
    fr_backward(g.contiguous(), b, s, 1, ref, overwrite=True)
    assert (bg.contiguous() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("points", [1, 5])
def test_autograd_channels_last(points):
    """feature_refine on a channels_last input: NHWC forward + NHWC backward, output and gradient channels_last,
    same values as the NCHW path."""
    from r3det.ops import feature_refine as M
    N, C, H, W, stride = 2, 32, 40, 24, 8
    r = np.random.default_rng(9)
    xv = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = dev(fr_boxes(N, H, W, stride, 4))
    top = dev(r.normal(size=(N, C, H, W)).astype(np.float32))
    calls = []
    real = M.fr_backward_nhwc
    M.fr_backward_nhwc = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        x1 = cl(xv).requires_grad_(True)
        y1 = M.feature_refine(x1, boxes, 1 / stride, points)
        assert y1.is_contiguous(memory_format=CL)
        y1.backward(top)
    finally:
        M.fr_backward_nhwc = real
    assert calls == [1] and x1.grad.is_contiguous(memory_format=CL)
    x2 = dev(xv).requires_grad_(True)
    y2 = M.feature_refine(x2, boxes, 1 / stride, points)
    y2.backward(top)
    assert torch.equal(y1.contiguous(), y2)
    assert (x1.grad.contiguous() - x2.grad).abs().max().item() <= 1e-5 * max(1.0, x2.grad.abs().max().item())


def test_module_training_channels_last():
    """FeatureRefineModule with grad on channels_last maps: gradients of the parameters and of the input equal the
    NCHW module's (same weights) within float tolerance, and the sampler ran on NHWC memory."""
    from r3det.ops import feature_refine as M
    torch.manual_seed(0)
    N, C = 2, 16
    strides = [8, 16]
    sizes = [(24, 24), (12, 12)]
    m1 = M.FeatureRefineModule(C, strides).cuda()
    m1.init_weights()
    for p in m1.parameters():
        torch.nn.init.normal_(p, 0, 0.2)
    m2 = M.FeatureRefineModule(C, strides).cuda()
    m2.load_state_dict(m1.state_dict())
    m1 = m1.to(memory_format=CL)
    xs = [torch.randn(N, C, h, w, device='cuda') for h, w in sizes]
    rois = [[dev(fr_boxes(1, h, w, s, 10 * i + s)) for (h, w), s in zip(sizes, strides)] for i in range(N)]
    # (the levels node -- round 5: the module's TAIL of both levels, add + samplers + residual add, as ONE channels_last
    # library call; the gradients come back channels_last from the _nhwc gathers)
    calls = []
    real = M.fr_module_levels_nhwc
    M.fr_module_levels_nhwc = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        x1 = [x.contiguous(memory_format=CL).requires_grad_(True) for x in xs]
        o1 = m1(x1, rois)
        sum((o * (i + 1)).sum() for i, o in enumerate(o1)).backward()
    finally:
        M.fr_module_levels_nhwc = real
    assert len(calls) == 1 and all(o.is_contiguous(memory_format=CL) for o in o1)
    assert all(x.grad.is_contiguous(memory_format=CL) for x in x1)
    x2 = [x.clone().requires_grad_(True) for x in xs]
    o2 = m2(x2, rois)
    sum((o * (i + 1)).sum() for i, o in enumerate(o2)).backward()
    for a, b in zip(o1, o2):
        assert torch.allclose(a.contiguous(), b, rtol=1e-4, atol=1e-4)
    for a, b in zip(x1, x2):
        assert torch.allclose(a.grad.contiguous(), b.grad, rtol=1e-4, atol=1e-4)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p1.grad.contiguous(), p2.grad, rtol=1e-3, atol=1e-3 * max(1.0, p2.grad.abs().max().item())), k
