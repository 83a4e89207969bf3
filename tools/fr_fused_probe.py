"""FeatureRefineModule tail at level 0 (N = 4, C = 256, 128 x 128): add + sampler + add as three launches
vs the fused sampler launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_forward_prepared, fr_module_prepared, fr_prepare  # noqa: E402

feats, boxes = syn.fr_pyramid(4, 256, 9, device="cuda")
for lvl in (0, 1):
    a = feats[lvl]
    b, res = torch.randn_like(a), torch.randn_like(a)
    N, C, H, W = a.shape
    table = fr_prepare(boxes[lvl], N, H, W, 1.0 / syn.STRIDES[lvl])
    out, tmp = torch.empty_like(a), torch.empty_like(a)

    def three():
        mixed = a + b
        fr_forward_prepared(mixed, table, tmp)
        return res + tmp

    def fused():
        fr_module_prepared(a, b, res, table, out)

    mixed = a + b

    def residual():
        fr_module_prepared(mixed, None, res, table, out)

    for name, fn in (("three launches", three), ("fused", fused), ("residual only", residual),
                     ("three launches", three), ("fused", fused), ("residual only", residual)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 20
        passes = 3 if name == "residual only" else 4
        print(f"level {lvl} {name:15s} {us:8.1f} us  ({4 * passes * a.numel() / us / 1e3:7.1f} GB/s on {passes - 1} reads + 1 write)",
              flush=True)
