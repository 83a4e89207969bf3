"""Round 6, VERDICT r5 next #3: the IoU matrix with fill and clip in ONE launch on disjoint addresses (option iou_impl 5:
K1 = the tests alone + survivor bits, K2 = the drain whose first workgroups write the zeros the bits do not name)
against the shipped stream (tests + zeros) -> drain pair (iou_impl 0).  Same process, alternating; outputs compared bit
for bit.  IOU_AB_NFILL: fill blocks of K2 (default: one per compute unit)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C  # noqa: E402
from r3det import synthetic as syn  # noqa: E402
from r3det.ops import obb_overlaps, rbbox_iou  # noqa: E402

dev = torch.device("cuda")
anchors = syn.anchor_grid(device=dev)
refined = torch.cat([syn.fr_level_boxes(1, 1024 // s, 1024 // s, s, 50 + i, device=dev) for i, s in enumerate(syn.STRIDES)])
shapes = (("128x196416", rbbox_iou, syn.dota_like_rboxes(128, 5, device=dev), anchors),
          ("512x196416", rbbox_iou, syn.dota_like_rboxes(512, 6, device=dev), anchors),
          ("128x21824", rbbox_iou, syn.dota_like_rboxes(128, 5, device=dev), refined),
          ("v3_128x196416", obb_overlaps, syn.dota_like_rboxes(128, 5, device=dev), anchors))
nfills = [int(x) for x in os.environ.get("IOU_AB_NFILL", "0").split(",")]


def timed(fn, b1, b2, reps=30):
    for _ in range(3):
        fn(b1, b2)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        out = fn(b1, b2)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps, out


for name, fn, b1, b2 in shapes:
    if os.environ.get("IOU_AB_SHAPE", name) != name:
        continue
    _C.set_option("iou_impl", 0)
    _, ref = timed(fn, b1, b2, 3)
    for rnd in range(3):
        line = [f"{name} round {rnd}:"]
        _C.set_option("iou_impl", 0)
        _C.set_option("iou_order", 8)
        us, _ = timed(fn, b1, b2)
        line.append(f"stream+drain {us:6.1f} us")
        for nf in nfills:
            _C.set_option("iou_impl", 5)
            _C.set_option("iou_nfill", nf)
            us5, out = timed(fn, b1, b2)
            line.append(f"| one launch (nfill {nf or 'CUs'}) {us5:6.1f} us {'==' if torch.equal(out, ref) else '!= (%d differ)' % int((out != ref).sum())}")
        print(" ".join(line), flush=True)
_C.set_option("iou_impl", 0)
_C.set_option("iou_order", 8)
