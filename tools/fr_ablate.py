"""Ablation of the persistent FR forward kernel (level 0, N=4, C=256): which phase costs what.
bits: 1 = no tap-table loads, 2 = no LDS gathers, 4 = no stores, 8 = no plane loads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch
from r3det import _C, synthetic as syn
from r3det.ops.feature_refine import fr_forward
dev = torch.device("cuda")
feats, boxes = syn.fr_pyramid(4, 256, 9, device=dev)
f, b = feats[0], boxes[0]
o = torch.empty_like(f)
_C.set_option("fr_impl", 5)
alg = 8 * f.numel() + 20 * b.size(0)
for dbg in (0, 1, 2, 3, 4, 8, 12, 7, 15):
    _C.set_option("fr_dbg", dbg)
    for _ in range(3):
        fr_forward(f, b, 1 / 8, 1, o)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fr_forward(f, b, 1 / 8, 1, o)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 100)
    ts.sort()
    print(f"dbg={dbg:2d} ({'notaps ' if dbg&1 else ''}{'nogather ' if dbg&2 else ''}{'nostore ' if dbg&4 else ''}{'noload' if dbg&8 else ''}) "
          f"{ts[3]:7.1f} us  {alg / ts[3] / 1e3:7.1f} GB/s-equivalent")
_C.set_option("fr_dbg", 0)
