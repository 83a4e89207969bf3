"""A/B inside one process: the NCHW pyramid pass with the coarse levels as ONE grid (default) against one launch per
level (option frb_impl 6): sampler forward (r3det_feature_refine_forward_levels) and the backward's gathers
(r3det_feature_refine_backward_levels_indexed), N = 4 and N = 2, C = 256, five levels of a 1024^2 input."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import feature_refine_levels, fr_forward_levels  # noqa: E402

dev = torch.device("cuda")
scales = [1.0 / s for s in syn.STRIDES]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / n


for N in (4, 2):
    feats, boxes = syn.fr_pyramid(N, 256, 9, device=dev)
    outs = [torch.empty_like(f) for f in feats]
    xs = [f.clone().requires_grad_(True) for f in feats]
    gs = [torch.randn_like(f) for f in feats]
    nbytes = sum(f.numel() for f in feats) * 8 + sum(b.numel() for b in boxes) * 4

    def fwd():
        fr_forward_levels(feats, boxes, scales, 1, outs)

    def fwd_bwd():
        for x in xs:
            x.grad = None
        torch.autograd.backward(feature_refine_levels(xs, boxes, scales, 1), gs)

    for rnd in range(2):
        for impl, what in ((6, "one launch per level"), (0, "coarse levels one grid")):
            _C.set_option("frb_impl", impl)
            t = timed(fwd)
            print(f"N={N} forward  5 levels, {what:24s}: {t:7.1f} us  ({nbytes / t / 8e6:.3f} of HBM peak)", flush=True)
            print(f"N={N} fwd+bwd  5 levels, {what:24s}: {timed(fwd_bwd):7.1f} us", flush=True)
_C.set_option("frb_impl", 0)
