"""rbbox_iou (v1) on the bench shapes, for rocprofv3 --kernel-trace: IOU_PROF_SHAPE = 128x196416 (assignment of
the base head), 128x21824 (refine stage), 1000x128 (BASELINE configs[0]) or all (default)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.ops import rbbox_iou  # noqa: E402

from r3det import _C  # noqa: E402

dev = torch.device("cuda")
for opt in ("iou_impl", "iou_qcap", "iou_small", "iou_dwgs", "clip_impl", "fr_walk", "iou_order", "iou_dyn"):  # A/B knobs, e.g. IOU_PROF_iou_qcap=100
    if os.environ.get("IOU_PROF_" + opt):
        _C.set_option(opt, int(os.environ["IOU_PROF_" + opt]))
which = os.environ.get("IOU_PROF_SHAPE", "all")
anchors = syn.anchor_grid(device=dev)
gt = syn.dota_like_rboxes(128, 5, device=dev)
refined = torch.cat([syn.fr_level_boxes(1, 1024 // s, 1024 // s, s, 50 + i, device=dev) for i, s in enumerate(syn.STRIDES)])
a, g = syn.rand_rboxes(1000, 0, device=dev), syn.rand_rboxes(128, 1, device=dev)
from r3det.ops import obb_overlaps  # noqa: E402

gt512 = syn.dota_like_rboxes(512, 6, device=dev)
if which == "vec":  # the aligned form (rbbox_geo_kernel.cu:271-309): 196 416 pairs, anchors against jittered copies
    jit = anchors.clone()
    jit[:, :2] += torch.randn_like(jit[:, :2]) * 4
    jit[:, 4] = -0.3
    for _ in range(3):
        rbbox_iou(anchors, jit, True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        out = rbbox_iou(anchors, jit, True)
    e.record()
    torch.cuda.synchronize()
    print(f"rbbox_iou vec 196416: {s.elapsed_time(e) * 1e3 / 20:8.1f} us per call  nnz {int((out > 0).sum())}", flush=True)
for name, b1, b2 in (("128x196416", gt, anchors), ("512x196416", gt512, anchors), ("128x21824", gt, refined),
                     ("1000x128", a, g), ("v3_128x196416", gt, anchors)):
    if which not in ("all", name):
        continue
    if name.startswith("v3"):
        rbbox_iou = obb_overlaps  # noqa: F811  (the v3 family: box_iou_rotated_ext.overlaps)
    for _ in range(3):
        rbbox_iou(b1, b2)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        out = rbbox_iou(b1, b2)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 20
    alg = 4 * out.numel() + 20 * (b1.size(0) + b2.size(0))
    print(f"rbbox_iou {name}: {us:8.1f} us per call  {alg / us / 1e3:8.1f} GB/s on {alg} algorithmic bytes  "
          f"nnz {int((out > 0).sum())}", flush=True)
