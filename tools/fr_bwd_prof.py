"""FR backward at the BASELINE training / inference shapes, both layouts, on rotating buffers (beyond the Infinity
Cache) -- the driver for `rocprofv3 --kernel-trace` / PMC passes of the backward kernels:
  NHWC: r3det_feature_refine_backward_nhwc (frb_index_kernel + frb_gather_kernel)
  NCHW: r3det_feature_refine_backward_ws  (frb_index_sort_kernel + frb_sell_kernel + frn_gather_kernel), and the
        gather alone on an index built ahead (r3det_feature_refine_backward_indexed: what a training step runs)
FR_BWD_LEVEL (0), FR_BWD_N (4), FR_BWD_FIELD=regular|adversarial|trained|grid."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import (fr_backward, fr_backward_index, fr_backward_indexed, fr_backward_nhwc,  # noqa: E402
                                      fr_backward_nhwc_index)

dev = torch.device("cuda")
level, N, C = int(os.environ.get("FR_BWD_LEVEL", 0)), int(os.environ.get("FR_BWD_N", 4)), 256
field = os.environ.get("FR_BWD_FIELD", "regular")
H = 128 >> level
stride = 8 << level
cl = torch.channels_last
boxes = syn.fr_level_boxes(N, H, H, stride, 3, device=dev)
if field == "adversarial":
    boxes[:, :2] = torch.rand(boxes.shape[0], 2, device=dev) * (H * stride)
elif field == "grid":  # no jitter at all: every box sits on its own cell's centre (what the gather's LDS reads cost without it)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(H, dtype=torch.float32, device=dev),
                            indexing='ij')
    boxes[:, 0] = (xs.reshape(-1) * stride).repeat(N)
    boxes[:, 1] = (ys.reshape(-1) * stride).repeat(N)
elif field == "trained":  # every position regresses to the centre of the object it lies on: piles of ~9-25 per cell
    g = (boxes[:, :2] / (4 * stride)).floor() * (4 * stride) + 2 * stride
    boxes[:, :2] = g + torch.randn_like(g) * 0.3 * stride
nset = max(3, int(0.9e9 // (2 * N * C * H * H * 4)))
sets_cl = [tuple(torch.randn(N, C, H, H, device=dev).contiguous(memory_format=cl) for _ in range(2)) for _ in range(nset)]
alg = 8 * N * C * H * H + 20 * N * H * H


def timed(name, fn, sets):
    for i in range(2 * len(sets)):
        fn(*sets[i % len(sets)])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    s.record()
    for i in range(reps):
        fn(*sets[i % len(sets)])
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / reps
    print(f"{name:34s} level {level} N={N} {field:11s}: {us:7.1f} us per call, {alg / us / 1e3:7.1f} GB/s on {alg} algorithmic bytes", flush=True)


timed("fr_backward_nhwc (index + gather)", lambda g, o: fr_backward_nhwc(g, boxes, 1.0 / stride, 1, o, overwrite=True), sets_cl)
ix = fr_backward_nhwc_index(boxes, N, H, H, 1.0 / stride, 1)
timed("fr_backward_nhwc (gather alone)", lambda g, o: fr_backward_nhwc(g, None, 1.0 / stride, 1, o, overwrite=True, index=ix), sets_cl)
del sets_cl
sets = [tuple(torch.randn(N, C, H, H, device=dev) for _ in range(2)) for _ in range(nset)]
timed("fr_backward NCHW (index + gather)", lambda g, o: fr_backward(g, boxes, 1.0 / stride, 1, o, overwrite=True), sets)
ix = fr_backward_index(boxes, N, C, H, H, 1.0 / stride, 1)
timed("fr_backward NCHW (gather alone)", lambda g, o: fr_backward_indexed(g, 1, o, ix), sets)

# ---- round 6: the index alone, from the box records against from the level's tap table (what a training step's forward
# launch leaves behind: r3det_feature_refine_*_levels_nhwc_tab / r3det_feature_refine_prepare)
from r3det.ops.feature_refine import (fr_backward_index_levels, fr_backward_nhwc_index_levels,  # noqa: E402
                                      fr_forward_levels_nhwc, tap_tables)

tabs = tap_tables(N, [(H, H)], dev)
f0 = torch.randn(N, 8, H, H, device=dev).contiguous(memory_format=cl)
assert fr_forward_levels_nhwc([f0], [boxes], [1.0 / stride], 1, [torch.empty_like(f0)], tabs)
one = [(None, None)]
for what, tb in (("boxes", None), ("tap table", tabs)):
    timed(f"index alone, CSR (nhwc), {what}", lambda a, b: fr_backward_nhwc_index_levels([boxes], N, [(H, H)], [1.0 / stride], 1, tb), one)
    timed(f"index alone, SELL (nchw), {what}", lambda a, b: fr_backward_index_levels([boxes], N, C, [(H, H)], [1.0 / stride], 1, tb), one)
