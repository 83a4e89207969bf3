"""The fused assignment (r3det_rbbox_assign) at 128 GT x 196 416 anchors, for rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
from r3det import synthetic as syn
from r3det.core.bbox.assigners import MaxIoUAssigner
dev = torch.device("cuda")
if os.environ.get("IOU_DWGS"):
    from r3det import _C
    _C.set_option("iou_dwgs", int(os.environ["IOU_DWGS"]))
if os.environ.get("CLIP_IMPL"):  # A/B: 1 = the LDS-list clip of rounds 2-4
    from r3det import _C
    _C.set_option("clip_impl", int(os.environ["CLIP_IMPL"]))
if os.environ.get("ASSIGN_PROBE"):  # probes build: what the drain emits (r3_iou.hip AssignOut.probe)
    from r3det import _C
    _C.set_option("fr_walk", 2000 + int(os.environ["ASSIGN_PROBE"]))
anchors = syn.anchor_grid(device=dev)
gt = syn.dota_like_rboxes(128, 5, device=dev)
a = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1, iou_calculator=dict(type='RBboxOverlaps2D_v1'))
for _ in range(3): a.assign(anchors, gt, None, None)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): r = a.assign(anchors, gt, None, None)
e.record(); torch.cuda.synchronize()
print(f"assign 128 x 196416: {s.elapsed_time(e) * 50:.1f} us per call, positives {int((r.gt_inds > 0).sum())}", flush=True)
