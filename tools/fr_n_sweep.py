"""FR forward level-0 timing for several batch sizes and kernel variants:
    python tools/fr_n_sweep.py [impl ...]   (default 10 2: cell, plane)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det import _C, synthetic as syn  # noqa: E402
from r3det.ops.feature_refine import fr_forward  # noqa: E402

impls = [int(a) for a in sys.argv[1:]] or [10, 2]
dev = torch.device("cuda")
for N in (1, 2, 4, 8, 16):
    feats, boxes = syn.fr_pyramid(N, 256, 9, device=dev)
    f, b = feats[0], boxes[0]
    o = torch.empty_like(f)
    alg = 8 * f.numel() + 20 * b.size(0)
    for impl in impls:
        _C.set_option("fr_impl", impl)
        for _ in range(5):
            fr_forward(f, b, 1 / 8, 1, o)
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                fr_forward(f, b, 1 / 8, 1, o)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3 / 20)
        ts.sort()
        print(f"N={N:2d} impl {impl}: med {ts[3]:7.1f} us  min {ts[0]:7.1f} us  {alg / ts[3] / 1e3:7.0f} GB/s", flush=True)
_C.set_option("fr_impl", 0)
