"""GPU parity: fused FR box producers (r3det_filter_bboxes) against the op-by-op torch form of
RRetinaHead.filter_bboxes / RRetinaRefineHead.refine_bboxes on the same device, for NCHW and
channels_last head outputs.  The best-anchor choice is an integer decision (exact); the decoded
boxes go through exp: tolerance 1e-6 relative (north_star: 1e-5), written below."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def heads():
    from r3det.models.heads import RRetinaHead, RRetinaRefineHead
    torch.manual_seed(3)
    return RRetinaHead().cuda().eval(), RRetinaRefineHead().cuda().eval()


def close(a, b):
    assert a.shape == b.shape
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-6), float((a - b).abs().max())


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("sizes", [[(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)], [(128, 128), (64, 64), (13, 7)]])
def test_filter_bboxes(channels_last, sizes):
    head, _ = heads()
    head.strides = head.strides[:len(sizes)]
    N, A, C = 3, head.num_anchors, head.cls_out_channels
    g = torch.Generator(device='cuda').manual_seed(5)
    cls = [torch.randn(N, A * C, h, w, device='cuda', generator=g) for h, w in sizes]
    reg = [torch.randn(N, A * 5, h, w, device='cuda', generator=g) * 0.5 for h, w in sizes]
    reg[0][0, 2::5] = 9.0    # dw far beyond the wh_ratio clip
    reg[0][1, 3::5] = -9.0
    cls[0][2, :, 0, 0] = 0.25  # a position where every anchor ties: the first one must win
    if channels_last:
        cls = [c.contiguous(memory_format=torch.channels_last) for c in cls]
        reg = [r.contiguous(memory_format=torch.channels_last) for r in reg]
    got = head.filter_bboxes(cls, reg)
    want = head.filter_bboxes_torch(cls, reg)
    for i in range(N):
        for l in range(len(sizes)):
            close(got[i][l], want[i][l])
    # the integer decision, checked on its own: decoded centre = anchor centre + anchor size * delta
    anc = head.anchors([c.shape[-2:] for c in cls], 'cuda')[0].reshape(-1, A, 5)
    c0 = cls[0].permute(0, 2, 3, 1).reshape(N, -1, A, C)
    best = c0.max(-1)[0].argmax(-1)
    assert best[2, 0] == 0
    aw = anc[torch.arange(anc.size(0)), best[2], 2]
    assert torch.allclose(got[2][0][:, 2] / aw, (reg[0].permute(0, 2, 3, 1).reshape(N, -1, A, 5)[2][
        torch.arange(anc.size(0)), best[2], 2]).clamp(-4.135166556742356, 4.135166556742356).exp(), rtol=1e-5)


@pytest.mark.parametrize("channels_last", [False, True])
def test_refine_bboxes(channels_last):
    _, ref = heads()
    sizes = [(32, 32), (16, 16), (5, 9)]
    ref.strides = ref.strides[:len(sizes)]
    N = 2
    g = torch.Generator(device='cuda').manual_seed(7)
    reg = [torch.randn(N, 5, h, w, device='cuda', generator=g) * 0.3 for h, w in sizes]
    cls = [torch.randn(N, 15, h, w, device='cuda', generator=g) for h, w in sizes]
    rois = [[torch.rand(h * w, 5, device='cuda', generator=g) * 100 + 1 for h, w in sizes] for _ in range(N)]
    if channels_last:
        reg = [r.contiguous(memory_format=torch.channels_last) for r in reg]
    got = ref.refine_bboxes(cls, reg, rois)
    want = ref.refine_bboxes_torch(cls, reg, rois)
    for i in range(N):
        for l in range(len(sizes)):
            close(got[i][l], want[i][l])


@pytest.mark.parametrize("A,C", [(3, 4), (2, 16), (9, 15), (5, 7)])
def test_filter_bboxes_layouts_agree(A, C):
    """The channels_last heads take the LDS-staged kernel (templated for 9 x 15, runtime loops otherwise,
    the strided kernel when a chunk of 64 positions is not 16-byte granular): same boxes as NCHW inputs."""
    from r3det.ops.fr_boxes import filter_bboxes
    N, H, W = 2, 24, 20
    g = torch.Generator(device='cuda').manual_seed(A * 31 + C)
    cls = torch.randn(N, A * C, H, W, device='cuda', generator=g)
    reg = torch.randn(N, A * 5, H, W, device='cuda', generator=g) * 0.3
    anchors = torch.rand(H * W * A, 5, device='cuda', generator=g) * 40 + 4
    want = filter_bboxes(cls, reg, anchors, A, C)
    got = filter_bboxes(cls.contiguous(memory_format=torch.channels_last),
                        reg.contiguous(memory_format=torch.channels_last), anchors, A, C)
    assert torch.equal(got, want)


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_producers_vs_reference_goldens(tag, channels_last):
    """r3det_filter_bboxes against outputs of the REFERENCE's own RRetinaHead.filter_bboxes /
    RRetinaRefineHead.refine_bboxes (tests/golden/heads.npz from make_golden_heads.py): random head maps
    with tied anchors at some positions and dw / dh beyond the wh_ratio clip.  The anchor choice is an
    integer decision: a wrong choice moves a box by whole anchor sizes, so the 1e-5 bar on the boxes
    (north_star) also pins it; the fused decode differs from torch's op-by-op form by rounding of exp."""
    import os

    import numpy as np

    from helpers import GOLDEN
    G = np.load(os.path.join(GOLDEN, "heads.npz"))
    head, ref = heads()
    fmt = torch.channels_last if channels_last else torch.contiguous_format

    def maps(prefix):
        return [torch.from_numpy(G[f"{tag}_{prefix}_l{l}"]).cuda().contiguous(memory_format=fmt) for l in range(5)]
    cls, reg = maps("cls"), maps("reg")
    assert float(max(r.abs().max() for r in reg)) > 4.2  # the clip is exercised
    rois = head.filter_bboxes(cls, reg)
    for i in range(2):
        for l in range(5):
            want = torch.from_numpy(G[f"{tag}_rois_{i}_l{l}"]).cuda()
            assert rois[i][l].shape == want.shape
            assert torch.allclose(rois[i][l], want, rtol=1e-5, atol=1e-5), (i, l, float((rois[i][l] - want).abs().max()))
    gold_rois = [[torch.from_numpy(G[f"{tag}_rois_{i}_l{l}"]).cuda() for l in range(5)] for i in range(2)]
    refined = ref.refine_bboxes(maps("rcls"), maps("rreg"), gold_rois)
    for i in range(2):
        for l in range(5):
            want = torch.from_numpy(G[f"{tag}_refined_{i}_l{l}"]).cuda()
            assert torch.allclose(refined[i][l], want, rtol=1e-5, atol=1e-5), (i, l)
