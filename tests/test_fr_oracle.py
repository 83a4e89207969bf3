"""CPU: the FR restatement has no reference fixture to pin it (the reference implements FR in
CUDA only and has no tests => "parity unpinned").  Mitigation: a second,
independent restatement in vectorised PyTorch must agree with the scalar C++ one, and the
backward must equal autograd of that PyTorch version."""
import numpy as np
import pytest
import torch

from helpers import fr_boxes
from oracle import api as O


def torch_fr(feat, boxes, scale, points):
    """Vectorised restatement of feature_refine_kernel.cu:16-65,112-163 (differentiable)."""
    N, C, H, W = feat.shape
    b = boxes.view(N, H * W, 5)
    roi_y = b[..., 0] * scale
    roi_x = b[..., 1] * scale
    pys, pxs = [roi_y], [roi_x]
    if points > 1:
        w_2, h_2 = b[..., 2] * scale / 2, b[..., 3] * scale / 2
        cosa, sina = torch.cos(b[..., 4]), torch.sin(b[..., 4])
        wx, wy, hx, hy = cosa * w_2, sina * w_2, -sina * h_2, cosa * h_2
        pxs += [roi_x + wx + hx, roi_x - wx + hx, roi_x - wx - hx, roi_x + wx - hx]
        pys += [roi_y + wy + hy, roi_y - wy + hy, roi_y - wy - hy, roi_y + wy - hy]
    flat = feat.reshape(N, C, H * W)
    out = flat.clone()
    for y, x in zip(pys, pxs):
        inside = ~((y < -1.0) | (y > H) | (x < -1.0) | (x > W))
        y = y.clamp(min=0)
        x = x.clamp(min=0)
        y_low, x_low = y.floor().long(), x.floor().long()
        ytop, xtop = y_low >= H - 1, x_low >= W - 1
        y_low = torch.where(ytop, torch.full_like(y_low, H - 1), y_low)
        x_low = torch.where(xtop, torch.full_like(x_low, W - 1), x_low)
        y_high = torch.where(ytop, y_low, y_low + 1)
        x_high = torch.where(xtop, x_low, x_low + 1)
        y = torch.where(ytop, y_low.to(y.dtype), y)
        x = torch.where(xtop, x_low.to(x.dtype), x)
        ly, lx = y - y_low, x - x_low
        hy, hx = 1 - ly, 1 - lx

        def g(yy, xx):
            idx = (yy * W + xx).clamp(0, H * W - 1)[:, None, :].expand(N, C, H * W)
            return flat.gather(2, idx)
        val = (hy * hx)[:, None] * g(y_low, x_low) + (hy * lx)[:, None] * g(y_low, x_high) + \
              (ly * hx)[:, None] * g(y_high, x_low) + (ly * lx)[:, None] * g(y_high, x_high)
        out = out + val * inside[:, None].to(val.dtype)
    return out.view(N, C, H, W)


@pytest.mark.parametrize("points", [1, 5])
@pytest.mark.parametrize("adversarial", [False, True])
def test_forward_two_restatements_agree(points, adversarial):
    N, C, H, W, stride = 2, 5, 12, 16, 8
    r = np.random.default_rng(0)
    feat = r.normal(size=(N, C, H, W)).astype(np.float32)
    boxes = fr_boxes(N, H, W, stride, 1, adversarial=adversarial)
    a = O.fr_forward(feat, boxes, 1 / stride, points)
    b = torch_fr(torch.from_numpy(feat).double(), torch.from_numpy(boxes).double(), 1 / stride, points)
    assert np.abs(a - b.numpy()).max() < 2e-5  # fp32 sampling coordinates vs fp64


@pytest.mark.parametrize("points", [1, 5])
def test_backward_matches_autograd(points):
    N, C, H, W, stride = 2, 3, 9, 9, 16
    r = np.random.default_rng(2)
    feat = torch.from_numpy(r.normal(size=(N, C, H, W))).double().requires_grad_(True)
    boxes = fr_boxes(N, H, W, stride, 3)
    top = r.normal(size=(N, C, H, W)).astype(np.float32)
    out = torch_fr(feat, torch.from_numpy(boxes).double(), 1 / stride, points)
    out.backward(torch.from_numpy(top).double())
    got = O.fr_backward(top, boxes, 1 / stride, points)
    assert np.abs(got - feat.grad.numpy()).max() < 5e-5


def test_xy_swap_quirk():
    """feature_refine_kernel.cu:131-132: row <- box[0] (x_ctr), column <- box[1] (y_ctr)."""
    feat = np.zeros((1, 1, 4, 6), np.float32)
    feat[0, 0, 3, 1] = 7.0  # row 3, col 1
    boxes = np.zeros((24, 5), np.float32)
    boxes[:, 0], boxes[:, 1] = 3.0, 1.0  # x_ctr = 3 -> ROW 3 ; y_ctr = 1 -> COL 1
    out = O.fr_forward(feat, boxes, 1.0, 1)
    assert np.allclose(out - feat, 7.0)


def test_border_and_out_of_range():
    feat = np.arange(12, dtype=np.float32).reshape(1, 1, 3, 4)
    def at(y, x):
        b = np.zeros((12, 5), np.float32)
        b[:, 0], b[:, 1] = y, x
        return (O.fr_forward(feat, b, 1.0, 1) - feat)[0, 0, 0, 0]
    assert at(-1.5, 0) == 0 and at(0, 4.5) == 0 and at(3.01, 0) == 0   # outside [-1, H] x [-1, W]
    assert at(-0.5, -0.5) == feat[0, 0, 0, 0]                           # clamped to 0
    assert at(2.7, 3.9) == feat[0, 0, 2, 3]                             # last row/col: frac = 0
    assert at(3.0, 4.0) == feat[0, 0, 2, 3]                             # y == H, x == W still sample
    assert at(0.5, 0.5) == pytest.approx(feat[0, 0, :2, :2].mean())
