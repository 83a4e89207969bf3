"""How many pairs pass the stream kernel's conservative tests (circumscribed circles, axis-aligned boxes), how many of
them overlap at all, and how many of the rest a separating-axis test on the two rectangles would catch -- for the IoU
shapes of the bench (recomputed with torch from the boxes; the kernels' tests are inflated by rounding margins).
    python tools/iou_survivors.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.ops import rbbox_iou  # noqa: E402

dev = torch.device("cuda")
anchors = syn.anchor_grid(device=dev)
gt = syn.dota_like_rboxes(128, 5, device=dev)
refined = torch.cat([syn.fr_level_boxes(1, 1024 // s, 1024 // s, s, 50 + i, device=dev) for i, s in enumerate(syn.STRIDES)])
a, g = syn.rand_rboxes(1000, 0, device=dev), syn.rand_rboxes(128, 1, device=dev)


def geo(b):
    x, y, w, h, t = b.unbind(1)
    c, s = torch.cos(t), torch.sin(t)
    ex = (w * c.abs() + h * s.abs()) / 2
    ey = (w * s.abs() + h * c.abs()) / 2
    return x, y, 0.5 * torch.sqrt(w * w + h * h), ex, ey, w / 2, h / 2, c, s


def sat_separated(A, B):
    """True where one of the four edge normals of the two rectangles separates them (exact arithmetic aside)."""
    ax, ay, _, _, _, aw, ah, ac, as_ = A
    bx, by, _, _, _, bw, bh, bc, bs = B
    dx, dy = bx[None, :] - ax[:, None], by[None, :] - ay[:, None]
    sep = torch.zeros_like(dx, dtype=torch.bool)
    for (ux, uy) in ((ac[:, None], as_[:, None]), (-as_[:, None], ac[:, None]), (bc[None, :], bs[None, :]), (-bs[None, :], bc[None, :])):
        d = (dx * ux + dy * uy).abs()
        ra = aw[:, None] * (ac[:, None] * ux + as_[:, None] * uy).abs() + ah[:, None] * (-as_[:, None] * ux + ac[:, None] * uy).abs()
        rb = bw[None, :] * (bc[None, :] * ux + bs[None, :] * uy).abs() + bh[None, :] * (-bs[None, :] * ux + bc[None, :] * uy).abs()
        sep |= d > ra + rb
    return sep


for name, b1, b2 in (("128x196416", gt, anchors), ("128x21824", gt, refined), ("1000x128", a, g)):
    A, B = geo(b1), geo(b2)
    dx, dy = A[0][:, None] - B[0][None, :], A[1][:, None] - B[1][None, :]
    surv = (dx * dx + dy * dy <= (A[2][:, None] + B[2][None, :]) ** 2) & (dx.abs() <= A[3][:, None] + B[3][None, :]) \
        & (dy.abs() <= A[4][:, None] + B[4][None, :])
    iou = rbbox_iou(b1, b2)
    over = iou > 0
    sat = sat_separated(A, B)
    n = b1.size(0) * b2.size(0)
    s = int(surv.sum())
    print(f"{name}: pairs {n}  survivors {s} ({100 * s / n:.2f} %)  overlapping {int(over.sum())} "
          f"({100 * int((over & surv).sum()) / max(1, s):.1f} % of survivors)  survivors a separating axis removes: "
          f"{int((surv & sat).sum())} ({100 * int((surv & sat).sum()) / max(1, s):.1f} %)  separated yet IoU > 0: {int((sat & over).sum())}",
          flush=True)
