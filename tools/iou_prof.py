"""rbbox_iou on the assignment shape (128 x 196416) and friends, for rocprofv3 --kernel-trace."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops import rbbox_iou  # noqa: E402

dev = torch.device("cuda")
anchors = syn.anchor_grid(device=dev)
for k in ((128,) if os.environ.get('IOU_PROF_128') else (128, 512)):
    gt = syn.dota_like_rboxes(k, 5, device=dev)
    for _ in range(3):
        rbbox_iou(gt, anchors)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        out = rbbox_iou(gt, anchors)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 20
    print(f"rbbox_iou {k}x{anchors.size(0)}: {us:8.1f} us  {out.numel() * 4 / us / 1e3:8.1f} GB/s  nnz {int((out > 0).sum())}", flush=True)
z = torch.empty(128 * anchors.size(0), device=dev)
for _ in range(3):
    z.zero_()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    z.zero_()
e.record()
torch.cuda.synchronize()
print(f"memset of the 128-row matrix: {s.elapsed_time(e) * 1e3 / 20:8.1f} us", flush=True)
