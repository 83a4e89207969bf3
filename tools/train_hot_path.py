"""Custom-op part of a TRAINING step (BASELINE configs[4] / SURVEY 8d config 5: batch 2 per GPU, 128 GT per
image): MaxIoU assignment of the 196 416 anchors and of the 21 824 refined boxes per image (fused, no overlap
matrix), FeatureRefine forward + backward over the pyramid (N = 2, C = 256).  Milliseconds per step, custom ops
only (no convolutions, no losses)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.core.bbox.assigners import MaxIoUAssigner  # noqa: E402
from r3det.ops.feature_refine import feature_refine, feature_refine_levels  # noqa: E402

dev = torch.device("cuda")
B, K = 2, 128
anchors = syn.anchor_grid(device=dev)
gts = [syn.dota_like_rboxes(K, 5 + i, device=dev) for i in range(B)]
feats, boxes = syn.fr_pyramid(B, 256, 9, device=dev)
refined = [torch.cat([b.view(B, -1, 5)[i] for b in boxes]) for i in range(B)]  # 21 824 boxes per image
asg1 = MaxIoUAssigner(0.5, 0.4, 0., iou_calculator=dict(type='RBboxOverlaps2D_v1'))
asg2 = MaxIoUAssigner(0.6, 0.5, 0., iou_calculator=dict(type='RBboxOverlaps2D_v1'))
xs = [f.clone().requires_grad_(True) for f in feats]
gs = [torch.randn_like(f) for f in feats]


def assign():
    for i in range(B):
        asg1.assign(anchors, gts[i])
        asg2.assign(refined[i], gts[i])


def fr_per_level():
    for x, b, g, s in zip(xs, boxes, gs, syn.STRIDES):
        x.grad = None
        feature_refine(x, b, 1.0 / s, 1).backward(g)


def fr():  # what FeatureRefineModule runs in NCHW training: the five levels as one autograd node
    for x in xs:
        x.grad = None
    torch.autograd.backward(feature_refine_levels(xs, boxes, [1.0 / s for s in syn.STRIDES], 1), gs)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n)
    return best * 1e3


ta, tf = timeit(assign), timeit(fr)
print(f"(one autograd node per level: {timeit(fr_per_level):7.3f} ms)")
from r3det import _C  # noqa: E402
for rnd in range(2):
    for impl, what in ((6, "index level by level"), (0, "indexes of all levels in one launch")):
        _C.set_option("frb_impl", impl)
        print(f"(levels node, {what}: {timeit(fr):7.3f} ms)")
_C.set_option("frb_impl", 0)
print(f"assignment ({B} x ({K} x {anchors.size(0)} + {K} x {refined[0].size(0)})): {ta:7.3f} ms")
print(f"FeatureRefine fwd + bwd, 5 levels (N = {B}, C = 256):                 {tf:7.3f} ms")
print(f"training hot path per step:                                           {ta + tf:7.3f} ms")
