"""FR forward level-1 (64 x 64) timing: cell kernel workgroup sizes (fr_dbg 1/2/3 = 1024/512/256
threads) against the plane kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_forward  # noqa: E402

dev = torch.device("cuda")
for N in (4, 16):
    feats, boxes = syn.fr_pyramid(N, 256, 9, device=dev)
    f, b = feats[1], boxes[1]
    o = torch.empty_like(f)
    ref = torch.empty_like(f)
    _C.set_option("fr_impl", 2)
    fr_forward(f, b, 1 / 16, 1, ref)
    alg = 8 * f.numel() + 20 * b.size(0)
    for impl, dbg in ((2, 0), (10, 1), (10, 2), (10, 3)):
        _C.set_option("fr_impl", impl)
        _C.set_option("fr_dbg", dbg)
        for _ in range(5):
            fr_forward(f, b, 1 / 16, 1, o)
        assert torch.equal(o, ref)
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                fr_forward(f, b, 1 / 16, 1, o)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3 / 20)
        ts.sort()
        print(f"N={N:2d} impl {impl} dbg {dbg}: med {ts[3]:7.1f} us  min {ts[0]:7.1f} us  {alg / ts[3] / 1e3:7.0f} GB/s", flush=True)
_C.set_option("fr_impl", 0)
_C.set_option("fr_dbg", 0)
