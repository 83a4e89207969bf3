"""points = 5 on channels_last memory, plain sampler per level (N = 4, C = 256): the tiled kernel (8 x 8 positions with
their sample region in LDS; default) against the simple kernel (fr_dbg 9 selects nothing for points 5 but switches the
tiled form off)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_forward_nhwc  # noqa: E402

dev = torch.device("cuda")
cl = torch.channels_last
N = int(os.environ.get("FR_N", 4))
feats, boxes = syn.fr_pyramid(N, 256, 9, device=dev)
for lvl in range(5):
    f, b = feats[lvl].contiguous(memory_format=cl), boxes[lvl]
    sets = [(torch.randn_like(f), torch.empty_like(f)) for _ in range(6 if lvl == 0 else 12)]
    ref = None
    for dbg in (9, 0):
        _C.set_option("fr_dbg", dbg)
        for x, o in sets:
            assert fr_forward_nhwc(x, b, 1 / syn.STRIDES[lvl], 5, o)
        torch.cuda.synchronize()
        if ref is None:
            ref = sets[0][1].clone()
        same = torch.equal(ref, sets[0][1])
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(24):
            x, o = sets[i % len(sets)]
            fr_forward_nhwc(x, b, 1 / syn.STRIDES[lvl], 5, o)
        e.record()
        torch.cuda.synchronize()
        nb = f.numel() * 8 + b.numel() * 4
        t = s.elapsed_time(e) * 1000 / 24
        print(f"points 5 nhwc level {lvl} N={N} {'simple' if dbg else 'tiled '}: {t:7.1f} us  ({nb / t / 8e6:.3f} of HBM peak)  "
              f"bit-identical to the simple kernel: {same}", flush=True)
    _C.set_option("fr_dbg", 0)
